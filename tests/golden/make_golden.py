#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the committed parity fixtures.

Run in the BUILD container (needs /root/reference and oracle/_ref built by
`make -f oracle/Makefile ref`); the fixtures it writes travel with the repo, the reference does not.

  reference_literals.json : golden literals lifted (by AST, values only) from the reference's own
                            test-suite -- the known-answer pins G1,G2,G8,G9,G11 of SURVEY.md 8(c).
  ref_outputs.npz         : outputs of the reference's own fknm/frne extension (oracle/_ref, built
                            unmodified from /root/reference) on seeded inputs -- the differential
                            pins for fkine / jacob0 / jacobe / hessian0 / ik_LM / rne.
  ref_python_ik.npz       : outputs of the reference's own PYTHON solvers (robot/IK.py: IK_LM / IK_NR / IK_GN
                            `.solve`, loaded unmodified under the stand-in modules of oracle/ref_python.py) with and
                            without the null-space terms kq, km, ps, pi, for explicit start-vector tables (so no RNG
                            is involved even across many searches), and of `fknm.Angle_Axis` on random, identical,
                            half-turn and near-threshold rotation pairs.  `python make_golden.py python_ik` writes
                            only this file.
"""
import ast
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_TESTS = "/root/reference/tests"
REF_ROOT = "/root/reference"


def literal(fname, func, var, nth=0):
    """Value of the nth `var = <expr>` inside `def func` of a reference test file."""
    src = open(os.path.join(REF_TESTS, fname)).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == func:
            hits = [a for a in ast.walk(node) if isinstance(a, ast.Assign)
                    and len(a.targets) == 1 and isinstance(a.targets[0], ast.Name)
                    and a.targets[0].id == var]
            hits.sort(key=lambda a: a.lineno)
            expr = ast.Expression(hits[nth].value)
            ast.fix_missing_locations(expr)
            val = eval(compile(expr, fname, "eval"), {"np": np, "pi": math.pi, "math": math})
            return np.asarray(val, dtype=float), hits[nth].lineno
    raise KeyError((fname, func, var))


def rodrigues(axis, th):
    a = np.asarray(axis, dtype=float)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * (K @ K)


PI_ARRAY = np.array([0.3, 0.2, 0.4, 0.25, 0.3, 0.5, 0.35])


def python_ik():
    """Fixtures from the reference's Python solvers and its compiled Angle_Axis (pins SURVEY rows a7 and a9)."""
    sys.path.insert(0, os.path.join(ROOT, "robotics-toolbox-python_amd"))
    import ctypes as C
    import rtbhip
    from oracle import chains, ref_python as rp
    out = {}
    rng = np.random.default_rng(20260922)

    # ---- Angle_Axis (core/fknm.cpp:112-162 -> _angle_axis core/ik.cpp:241-286)
    def se3(R, t):
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
        return T
    Te, Tep, tag = [], [], []
    def add(a, b, what):
        Te.append(a); Tep.append(b); tag.append(what)
    for _ in range(64):
        add(se3(rodrigues(rng.normal(size=3), rng.uniform(-3.1, 3.1)), rng.normal(size=3)),
            se3(rodrigues(rng.normal(size=3), rng.uniform(-3.1, 3.1)), rng.normal(size=3)), 0)          # generic: atan2 branch
    for _ in range(6):
        R = rodrigues(rng.normal(size=3), rng.uniform(-3, 3))
        add(se3(R, rng.normal(size=3)), se3(R, rng.normal(size=3)), 1)                                   # identical: |li| = 0, tr = 3
    for ax in ([1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, -2, 3]):
        R0 = rodrigues(rng.normal(size=3), rng.uniform(-3, 3))
        add(se3(R0, rng.normal(size=3)), se3(rodrigues(ax, math.pi) @ R0, rng.normal(size=3)), 2)        # half turn: |li| ~ 0, tr = -1
        add(se3(np.eye(3), np.zeros(3)), se3(rodrigues(ax, math.pi), np.zeros(3)), 2)
    for th in (4e-7, 4.9e-7, 5.1e-7, 6e-7, 1e-5, math.pi - 4e-7, math.pi - 4.9e-7, math.pi - 5.1e-7, math.pi - 6e-7, math.pi - 1e-5):
        for ax in ([0, 0, 1], [1, 2, -1]):                                                               # |li| = 2 sin(th) around the 1e-6 test
            R0 = rodrigues(rng.normal(size=3), rng.uniform(-3, 3))
            add(se3(R0, rng.normal(size=3)), se3(rodrigues(ax, th) @ R0, rng.normal(size=3)), 3)
    out["aa_Te"], out["aa_Tep"], out["aa_tag"] = np.array(Te), np.array(Tep), np.array(tag)
    out["aa_e"] = np.array([rp.angle_axis(a, b) for a, b in zip(Te, Tep)])

    # ---- IK_LM / IK_NR / IK_GN .solve (robot/IK.py:174-367, 507-576, 736-763, 994-1017, 1176-1220)
    ch = chains.panda_ets(with_limits=True)
    duck = rp.DuckETS(ch)
    ets = rtbhip.models.Panda().ets()
    ets.qlim = chains.PANDA_QLIM
    SEED, N, SMAX = 1234, 16, 100
    lib, h = rtbhip.lib(), ets._handle()
    starts = np.empty((N, SMAX, 7))             # the device generator's start vectors (host function of librtbhip, no GPU needed)
    buf = np.empty(7)
    for i in range(N):
        for d in range(SMAX):
            assert lib.rtbhip_ik_restart(h, SEED, i, d, buf.ctypes.data_as(C.c_void_p)) == 0
            starts[i, d] = buf
    out["ik_seed"], out["ik_starts"] = np.array(SEED), starts
    qs = rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7))
    Tep = np.array([duck.eval(x) for x in qs])
    Tep[5, :3, 3] += 2.5                         # one unreachable target: every search runs to ilimit
    out["ik_Tep"] = Tep
    q0 = qs + 0.2 * rng.normal(size=qs.shape)
    out["ik_q0"] = q0
    # near-solution, near-limit problems for the null-space terms (as tests/test_kernel_emu.py's emu check)
    qn = rng.uniform(ch.qlim[0] + 0.25, ch.qlim[1] - 0.25, (N, 7))
    qn[::2, 3] = ch.qlim[1, 3] - 0.12
    qn[1::4, 1] = ch.qlim[0, 1] + 0.1
    Tn = np.array([duck.eval(x) for x in qn])
    q0n = np.clip(qn + 0.03 * rng.normal(size=qn.shape), ch.qlim[0] + 0.02, ch.qlim[1] - 0.02)
    out["ikn_Tep"], out["ikn_q0"] = Tn, q0n

    def run(key, solver, T, first, slimit, **kw):
        res = []
        for i in range(N):
            tab = starts[i, :slimit].copy()
            if first is not None:
                tab[0] = first[i]
            res.append(rp.solve(solver, duck, T[i], tab, slimit=slimit, **kw))
        out[key + "_q"] = np.array([r[0] for r in res])
        out[key + "_meta"] = np.array([[r[1], r[2], r[3]] for r in res], dtype=np.int64)
        out[key + "_E"] = np.array([r[4] for r in res])
        print("  %-16s success %2d/%d  iterations %5d  searches %4d" % (key, sum(r[1] for r in res), N, sum(r[2] for r in res), sum(r[3] for r in res)))

    run("lm_chan", "IK_LM", Tep, None, 100, method="chan", k=1.0)
    run("lm_wampler", "IK_LM", Tep, None, 100, method="wampler", k=0.01)
    run("lm_sugihara", "IK_LM", Tep, None, 100, method="sugihara", k=0.01)
    run("lm_chan_q0", "IK_LM", Tep, q0, 100, method="chan", k=1.0)
    run("lm_chan_nojl_mask", "IK_LM", Tep, None, 40, method="chan", k=0.1, joint_limits=False, mask=[1, 1, 1, 0.5, 0.5, 0.25])
    run("lm_chan_short", "IK_LM", Tep, None, 4, method="chan", k=1.0, ilimit=5)
    run("nr_q0", "IK_NR", Tn, q0n, 5, pinv=True)
    run("gn_q0", "IK_GN", Tn, q0n, 5, pinv=True)
    run("lm_chan_ns", "IK_LM", Tn, q0n, 3, method="chan", k=1.0, kq=0.1, km=0.1, ps=0.0, pi=0.3)         # tests/test_IK.py:194-196
    run("lm_sugihara_ns", "IK_LM", Tn, q0n, 3, method="sugihara", k=0.01, kq=0.5, km=0.0, ps=0.05, pi=0.4)
    run("lm_wampler_ns_km", "IK_LM", Tn, q0n, 3, method="wampler", k=0.01, kq=0.0, km=0.5)               # km alone: the guard at IK.py:572 drops it
    run("nr_ns", "IK_NR", Tn, q0n, 3, pinv=True, kq=0.01, km=1.0)                                        # tests/test_IK.py:166-173
    run("gn_ns", "IK_GN", Tn, q0n, 3, pinv=True, kq=1.0, km=1.0)                                         # tests/test_IK.py:261-263
    # ---- IK_QP .solve (robot/IK.py:1222-1520): the reference's own step code; the absent qpsolvers/quadprog is stood in for by
    # the exact enumerating solver of oracle/qp.py (the minimiser of a strictly convex QP is unique)
    run("qp_class_default", "IK_QP", Tep, None, 100)                                                      # IK_QP(): kj = 0.01, ks = 1
    run("qp_ets_default_q0", "IK_QP", Tn, q0n, 5, kj=1.0, ks=1.0)                                          # ETS.ikine_QP defaults (ETS.py:2942)
    run("qp_km", "IK_QP", Tn, q0n, 3, kj=0.1, ks=1.0, km=10.0)
    run("qp_mask_nojl", "IK_QP", Tep, q0, 20, kj=0.1, ks=2.0, joint_limits=False, mask=[1, 1, 1, 0.5, 0.5, 0.25])
    # kq > 0: the velocity dampers become inequality rows (IK.py:1453-1481); near-limit problems so rows are active
    run("qp_kq", "IK_QP", Tn, q0n, 3, kj=0.01, ks=1.0, kq=1.0, ps=0.0, pi=0.3)
    run("qp_kq_km", "IK_QP", Tn, q0n, 3, kj=0.1, ks=1.0, kq=0.5, km=10.0, ps=0.05, pi=0.4)
    run("qp_kq_far", "IK_QP", Tep, None, 30, kj=0.01, ks=1.0, kq=2.0, ps=0.0, pi=0.3)
    # an influence distance PER JOINT (IK.py:519-520 and :1441-1442 accept an ndarray)
    run("lm_chan_ns_pi_array", "IK_LM", Tn, q0n, 3, method="chan", k=1.0, kq=0.1, km=0.1, ps=0.0, pi=PI_ARRAY)
    run("qp_kq_pi_array", "IK_QP", Tn, q0n, 3, kj=0.01, ks=1.0, kq=1.0, ps=0.02, pi=PI_ARRAY)
    path = os.path.join(HERE, "ref_python_ik.npz")
    np.savez_compressed(path, **out)
    print("wrote ref_python_ik.npz with", len(out), "arrays,", os.path.getsize(path), "bytes")


def xacro_golden():
    """tests/golden/xacro/demo_arm.expected.urdf: the synthetic description of tests/golden/xacro (written for this repository) expanded by the
    REFERENCE's own xacro tool (tools/xacro/__init__.py, stdlib only) -- what rtbhip.xacro must reproduce element for element."""
    from pathlib import PurePosixPath
    sys.path.append(os.path.join(REF_ROOT, "src", "roboticstoolbox", "tools"))      # appended: the folder also holds a types.py
    import xacro
    src = PurePosixPath(os.path.join(HERE, "xacro", "demo_description", "urdf", "demo_arm.urdf.xacro"))
    out = xacro.main(src, None)
    dst = os.path.join(HERE, "xacro", "demo_arm.expected.urdf")
    open(dst, "w").write(out)
    print("wrote", dst, len(out), "bytes")


def main():
    if sys.argv[1:] == ["xacro"]:
        xacro_golden()
        return
    if sys.argv[1:] == ["python_ik"]:
        return python_ik()
    lit = {}

    def add(key, fname, func, var, nth=0):
        val, line = literal(fname, func, var, nth)
        lit[key] = {"source": "tests/%s:%d (%s)" % (fname, line, func), "value": val.tolist()}

    # G1 / G2 / G8 -- ETS Panda at q = [1.4,0.2,1.8,0.7,0.1,3.1,2.9]
    add("panda_q", "test_Robot.py", "test_fkine", "q1")
    add("G1_panda_fkine", "test_Robot.py", "test_fkine", "ans")
    add("G2_panda_jacob0", "test_ETS.py", "test_jacob0_panda", "ans")
    add("G2_panda_jacob0_robot", "test_Robot.py", "test_jacob0", "ans")
    add("G8_panda_hessian0_raw", "test_ETS.py", "test_hessian0", "ans")            # stored [:, :, i]
    add("G8_panda_hessian0_tool_raw", "test_ETS.py", "test_hessian0_tool", "ans")  # chain w/o ee, tool=ee
    # G9 -- Puma560 rne at qn
    for k in range(6):
        add("G9_puma_rne_tr%d" % k, "test_DHRobot.py", "test_rne", "tr%d" % k)
    add("G9_fext", "test_DHRobot.py", "test_rne", "fext")
    # Dynamics-mixin goldens: Puma560 at qn
    add("D_puma_accel_qd", "test_DHRobot.py", "test_accel", "qd")
    add("D_puma_accel_torque", "test_DHRobot.py", "test_accel", "torque")
    add("D_puma_accel", "test_DHRobot.py", "test_accel", "res")
    add("D_puma_inertia", "test_DHRobot.py", "test_inertia", "Ir")
    add("D_puma_coriolis_qd", "test_DHRobot.py", "test_coriolis", "qd")
    add("D_puma_coriolis", "test_DHRobot.py", "test_coriolis", "Cr")
    add("D_puma_gravload", "test_DHRobot.py", "test_gravload", "taur")
    add("D_puma_itorque_qdd", "test_DHRobot.py", "test_itorque", "qdd")
    add("D_puma_itorque", "test_DHRobot.py", "test_itorque", "tauir")
    # manipulability / jacobm goldens
    add("K_panda_jacobm_q", "test_ERobot.py", "test_jacobm", "q1")
    add("K_panda_jacobm", "test_ERobot.py", "test_jacobm", "ans")
    add("K_panda_partial_fkine3", "test_ETS.py", "test_partial_fkine", "ans")     # (7,7,6,7) at panda_q
    # G11 -- DH robots
    add("G11_dh_rprp_fkine", "test_DHRobot.py", "test_fkine", "T1")
    add("G11_dh_panda_fkine", "test_DHRobot.py", "test_fkine_panda", "T")
    add("G11_dh_rprp_jacobe", "test_DHRobot.py", "test_jacobe", "Je")
    for k in range(1, 8):
        add("G11_dh_panda_t%d" % k, "test_DHRobot.py", "test_fkine_all", "t%d" % k)
    with open(os.path.join(HERE, "reference_literals.json"), "w") as f:
        json.dump(lit, f, indent=0)
    print("wrote reference_literals.json with", len(lit), "entries")

    # ------------------------------------------------------------------ reference-run outputs
    from oracle import chains, ref_harness as rh
    out = {}
    rng = np.random.default_rng(20240922)
    panda = chains.panda_ets(with_limits=True)
    ref = rh.RefETS(panda)
    q = rng.uniform(-np.pi, np.pi, (64, 7))
    tool = chains.elementary("tx", 0.1) @ chains.elementary("Ry", 0.3) @ chains.elementary("tz", -0.05)
    base = chains.elementary("Rz", 0.7) @ chains.elementary("tx", 0.2) @ chains.elementary("Rx", -0.4)
    out["panda_q"] = q
    out["panda_tool"] = tool
    out["panda_base"] = base
    out["panda_fkine"] = ref.fkine(q)
    out["panda_fkine_bt"] = ref.fkine(q, base=np.asfortranarray(base), tool=np.asfortranarray(tool))
    out["panda_jacob0"] = ref.jacob0_batch(q)
    out["panda_jacobe"] = ref.jacobe_batch(q)
    out["panda_jacob0_tool"] = ref.jacob0_batch(q, tool=np.asfortranarray(tool))
    out["panda_jacobe_tool"] = ref.jacobe_batch(q, tool=np.asfortranarray(tool))
    out["panda_hessian0"] = np.array([ref.hessian0(q[i]) for i in range(8)])

    # a chain exercising every axis kind, flips, an arbitrary SE3 constant and non-monotone order
    mixed = chains.Chain([
        ("Rx", None, True), ("tx", 0.3), ("ty", None), ("Ry", 0.4), ("Ry", None),
        chains.elementary("Rz", 0.3) @ chains.elementary("tx", 0.2) @ chains.elementary("Rx", 1.1),
        ("tz", None, True), ("Rz", None), ("tx", None), ("Rx", -0.7), ("Ry", None, True), ("tz", 0.25),
    ], name="mixed")
    refm = rh.RefETS(mixed)
    qm = rng.uniform(-2, 2, (32, mixed.n))
    out["mixed_q"] = qm
    out["mixed_fkine"] = refm.fkine(qm)
    out["mixed_jacob0"] = refm.jacob0_batch(qm)
    out["mixed_jacobe"] = refm.jacobe_batch(qm)

    # DH robots lowered to ETS (config 1 plumbing) + dynamics
    puma = chains.puma560()
    refp = rh.RefETS(puma.ets())
    qp = rng.uniform(puma.qlim[:, 0], puma.qlim[:, 1], (32, 6))
    out["puma_q"] = qp
    out["puma_fkine"] = refp.fkine(qp)
    out["puma_jacob0"] = refp.jacob0_batch(qp)
    rne = rh.RefRNE(puma.L24(), 0)
    qd, qdd = rng.normal(size=(32, 6)), rng.normal(size=(32, 6))
    out["puma_qd"], out["puma_qdd"] = qd, qdd
    out["puma_rne"] = rne.rne(qp, qd, qdd)
    out["puma_rne_fext"] = rne.rne(qp, qd, qdd, fext=[1, 2, 3, 1, 2, 3])
    out["puma_rne_g0"] = rne.rne(qp, qd, qdd, gravity=[0, 0, 0])
    out["puma_rne_gx"] = rne.rne(qp, qd, qdd, gravity=[1.5, -2.0, -9.0])

    pdh = chains.panda_dh()
    refd = rh.RefETS(pdh.ets())
    qa = rng.uniform(pdh.qlim[:, 0], pdh.qlim[:, 1], (32, 7))
    out["pandadh_q"] = qa
    out["pandadh_fkine"] = refd.fkine(qa)
    rne2 = rh.RefRNE(pdh.L24(), 1)
    qd, qdd = rng.normal(size=(32, 7)), rng.normal(size=(32, 7))
    out["pandadh_qd"], out["pandadh_qdd"] = qd, qdd
    out["pandadh_rne"] = rne2.rne(qa, qd, qdd)
    out["pandadh_rne_fext"] = rne2.rne(qa, qd, qdd, fext=[-1, 0.5, 2, 0.3, -0.2, 0.1])

    # prismatic-containing DH / MDH chains for the RNE branches not hit by the two arms above
    for mdh in (0, 1):
        rows = [[0.3, 0.1, 0.0, 0.2, 0, 0.1, 0], [-0.4, 0.2, 0.5, 0.0, 1, 0.05, 0],
                [1.1, 0.0, 0.0, 0.15, 0, 0, 0], [0.6, -0.1, -0.3, 0.0, 1, 0, 0]]
        dyn = [(1.0 + i, [0.01 * i, -0.02, 0.03], [0.1, 0.2, 0.3, 0.01, -0.02, 0.03], 1e-4, 10.0 - 3 * i,
                1e-3, [0.1, -0.2]) for i in range(4)]
        tab = chains.DHTable("rprp%d" % mdh, mdh, rows, dyn)
        r4 = rh.RefRNE(tab.L24(), mdh)
        q4, qd4, qdd4 = rng.normal(size=(16, 4)), rng.normal(size=(16, 4)), rng.normal(size=(16, 4))
        out["rprp%d_q" % mdh], out["rprp%d_qd" % mdh], out["rprp%d_qdd" % mdh] = q4, qd4, qdd4
        out["rprp%d_L24" % mdh] = tab.L24()
        out["rprp%d_rne" % mdh] = r4.rne(q4, qd4, qdd4, gravity=[0.5, -1.0, -9.81], fext=[1, 2, 3, 4, 5, 6])
    # prismatic-first chains (the j==0 prismatic branches, ne.c:187-205 and :294-313)
    for mdh in (0, 1):
        rows = [[0.2, 0.1, 0.3, 0.0, 1, 0.0, 0], [0.7, 0.2, 0.0, 0.1, 0, 0, 0], [-0.5, 0.0, 0.2, 0.0, 1, 0.1, 0]]
        dyn = [(2.0 + i, [0.02, 0.01 * i, -0.03], [0.2, 0.1, 0.3, 0.0, 0.01, 0.0], 2e-4, 5.0, 2e-3, [0.05, -0.04])
               for i in range(3)]
        tab = chains.DHTable("prp%d" % mdh, mdh, rows, dyn)
        r3 = rh.RefRNE(tab.L24(), mdh)
        q3, qd3, qdd3 = rng.normal(size=(16, 3)), rng.normal(size=(16, 3)), rng.normal(size=(16, 3))
        out["prp%d_q" % mdh], out["prp%d_qd" % mdh], out["prp%d_qdd" % mdh] = q3, qd3, qdd3
        out["prp%d_L24" % mdh] = tab.L24()
        out["prp%d_rne" % mdh] = r3.rne(q3, qd3, qdd3, gravity=[0.5, -1.0, -9.81])

    # IK: supplied q0 near the solution (first search converges => no RNG involved)
    qs = rng.uniform(panda.qlim[0], panda.qlim[1], (24, 7))
    Tep = ref.fkine(qs)
    q0 = qs + 0.1 * rng.normal(size=qs.shape)
    out["ik_Tep"], out["ik_q0"] = Tep, q0
    for meth, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):
        res = [ref.ik_LM(Tep[i], q0=q0[i], k=k, method=meth) for i in range(len(qs))]
        out["ik_%s_q" % meth] = np.array([r[0] for r in res])
        out["ik_%s_meta" % meth] = np.array([[r[1], r[2], r[3]] for r in res], dtype=np.int64)
        out["ik_%s_E" % meth] = np.array([r[4] for r in res])
    # Gauss-Newton / Newton-Raphson (IK_GN_c / IK_NR_c), same targets and starts; + a 6-joint arm without pinv
    for meth, fn in (("gn", ref.ik_GN), ("nr", ref.ik_NR)):
        res = [fn(Tep[i], q0=q0[i]) for i in range(len(qs))]
        out["ik_%s_q" % meth] = np.array([r[0] for r in res])
        out["ik_%s_meta" % meth] = np.array([[r[1], r[2], r[3]] for r in res], dtype=np.int64)
        out["ik_%s_E" % meth] = np.array([r[4] for r in res])
    res = [ref.ik_NR(Tep[i], q0=q0[i], pinv_damping=0.05) for i in range(len(qs))]
    out["ik_nrd_q"] = np.array([r[0] for r in res])
    out["ik_nrd_meta"] = np.array([[r[1], r[2], r[3]] for r in res], dtype=np.int64)
    pets = puma.ets()
    pets.qlim = np.array([puma.qlim[:, 0], puma.qlim[:, 1]])
    refp6 = rh.RefETS(pets)
    qs6 = rng.uniform(puma.qlim[:, 0], puma.qlim[:, 1], (16, 6))
    T6 = refp6.fkine(qs6)
    q06 = qs6 + 0.05 * rng.normal(size=qs6.shape)
    out["ik6_Tep"], out["ik6_q0"] = T6, q06
    for meth, fn in (("gn", refp6.ik_GN), ("nr", refp6.ik_NR)):
        res = [fn(T6[i], q0=q06[i], pinv=False) for i in range(len(qs6))]
        out["ik6_%s_q" % meth] = np.array([r[0] for r in res])
        out["ik6_%s_meta" % meth] = np.array([[r[1], r[2], r[3]] for r in res], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "ref_outputs.npz"), **out)
    print("wrote ref_outputs.npz with", len(out), "arrays,",
          os.path.getsize(os.path.join(HERE, "ref_outputs.npz")), "bytes")
    python_ik()


if __name__ == "__main__":
    main()

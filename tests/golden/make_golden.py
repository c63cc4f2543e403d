#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the committed parity fixtures.

Run in the BUILD container (needs /root/reference and oracle/_ref built by
`make -f oracle/Makefile ref`); the fixtures it writes travel with the repo, the reference does not.

  reference_literals.json : golden literals lifted (by AST, values only) from the reference's own
                            test-suite -- the known-answer pins G1,G2,G8,G9,G11 of SURVEY.md 8(c).
  ref_outputs.npz         : outputs of the reference's own fknm/frne extension (oracle/_ref, built
                            unmodified from /root/reference) on seeded inputs -- the differential
                            pins for fkine / jacob0 / jacobe / hessian0 / ik_LM / rne.
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


def main():
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


if __name__ == "__main__":
    main()

"""-m "not gpu": pins the CPU oracle (oracle/rtb_oracle.c) against
  (a) the golden literals of the reference's own tests (tests/golden/reference_literals.json),
  (b) outputs of the reference's own fknm/frne build on seeded inputs (tests/golden/ref_outputs.npz),
  (c) the live oracle/_ref build when it is present (build container)."""
import numpy as np
import numpy.testing as nt
import pytest

from oracle import oracle, chains, ref_harness
from helpers import literals, ref_outputs, mixed_spec, tool_base, ref_python_ik, PY_IK_CASES, py_ik_problem, angle_axis_tolerance

LIT = literals()
REF = ref_outputs()
PY = ref_python_ik()


def test_G1_G2_panda_fkine_jacob0_literals():
    ch = chains.panda_ets()
    q = LIT["panda_q"]
    nt.assert_array_almost_equal(oracle.fkine(ch, q)[0], LIT["G1_panda_fkine"], decimal=6)   # test_Robot.py:18-33
    nt.assert_array_almost_equal(oracle.jacob0(ch, q)[0], LIT["G2_panda_jacob0"], decimal=6)  # test_ETS.py:262-363
    nt.assert_array_almost_equal(oracle.jacob0(ch, q)[0], LIT["G2_panda_jacob0_robot"], decimal=6)


def test_G3_jacobe_is_rotated_jacob0():
    ch = chains.panda_ets()
    q = LIT["panda_q"]
    T = oracle.fkine(ch, q)[0]
    R = T[:3, :3]
    tr2jac = np.zeros((6, 6)); tr2jac[:3, :3] = R.T; tr2jac[3:, 3:] = R.T   # test_ETS.py:365-398
    nt.assert_array_almost_equal(oracle.jacobe(ch, q)[0], tr2jac @ oracle.jacob0(ch, q)[0], decimal=12)


def test_G8_panda_hessian_literals():
    ch = chains.panda_ets()
    q = LIT["panda_q"]
    raw = LIT["G8_panda_hessian0_raw"]                      # stored as [:, :, i]; test_ETS.py:1118-1127
    ans = np.stack([raw[:, :, i] for i in range(7)])
    nt.assert_array_almost_equal(oracle.hessian0(ch, q)[0], ans, decimal=6)
    # with the ee segment moved into `tool` (test_ETS.py:1130-1579)
    arm = chains.Chain(chains.PANDA_ETS[:-2])
    ee = chains.elementary("tz", 103 * 1e-3) @ chains.elementary("Rz", -np.pi / 4)
    raw = LIT["G8_panda_hessian0_tool_raw"]
    ans = np.stack([raw[:, :, i] for i in range(7)])
    nt.assert_array_almost_equal(oracle.hessian0(arm, q, tool=ee)[0], ans, decimal=6)


def test_G9_puma_rne_literals():
    pu = chains.puma560()
    z, o = np.zeros(6), np.ones(6)
    g = -pu.gravity                                        # what frne.frne receives (DHRobot.py:1449)
    L = pu.L24()
    qn = chains.PUMA_QN
    cases = [(z, z, g, None), (z, o, g, None), (o, o, g, None), (o, z, g, None), (o, o, np.zeros(3), None),
             (z, z, g, LIT["G9_fext"])]
    for k, (qd, qdd, gg, fext) in enumerate(cases):        # test_DHRobot.py:1036-1062
        tau = oracle.rne_dh(L, 0, qn, qd, qdd, gg, fext)[0]
        nt.assert_array_almost_equal(tau, LIT["G9_puma_rne_tr%d" % k], decimal=4)


def test_G11_dh_literals():
    # RP-RP robot (test_DHRobot.py:170-189): PrismaticDH(), RevoluteDH(), PrismaticDH(theta=2), RevoluteDH()
    rows = [[0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0], [0, 0, 2.0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0]]
    T = oracle.dh_fkine(rows, 0, [1, 2, 3, 4])[0]
    nt.assert_array_almost_equal(T, LIT["G11_dh_rprp_fkine"], decimal=6)
    # ... and through the DH->ETS lowering (two independent reference formulations must agree)
    tab = chains.DHTable("rprp", 0, rows)
    nt.assert_array_almost_equal(oracle.fkine(tab.ets(), [1, 2, 3, 4])[0], T, decimal=12)
    # DH Panda at q = 1..7 (test_DHRobot.py:438-451, 4 dp)
    pd = chains.panda_dh()
    q = np.arange(1, 8, dtype=float)
    Tp = oracle.dh_fkine(pd.dh, 1, q, tool=pd.tool)[0]
    nt.assert_array_almost_equal(Tp, LIT["G11_dh_panda_fkine"], decimal=4)
    nt.assert_array_almost_equal(oracle.fkine(pd.ets(), q)[0], Tp, decimal=12)
    # all intermediate frames (test_DHRobot.py:638-710)
    for k in range(1, 8):
        Tk = oracle.dh_fkine(pd.dh[:k], 1, q[:k])[0]
        nt.assert_array_almost_equal(Tk, LIT["G11_dh_panda_t%d" % k], decimal=4)
    # jacobe of PrismaticDH(theta=4), RevoluteDH(a=2), PrismaticDH(theta=2), RevoluteDH() (test_DHRobot.py:453-479)
    rows = [[0, 0, 4.0, 0, 1, 0, 0], [0, 2.0, 0, 0, 0, 0, 0], [0, 0, 2.0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0]]
    tab = chains.DHTable("prpr", 0, rows)
    nt.assert_array_almost_equal(oracle.jacobe(tab.ets(), [1, 2, 3, 4])[0], LIT["G11_dh_rprp_jacobe"], decimal=4)


def test_config1_puma_fkine_closed_form_equals_ets_lowering():
    """BASELINE configs[0]: Puma560 6-DOF DH, fkine over 1e3 random q on the CPU path (plumbing)."""
    pu = chains.puma560()
    rng = np.random.default_rng(0)
    q = rng.uniform(pu.qlim[:, 0], pu.qlim[:, 1], (1000, 6))
    nt.assert_allclose(oracle.fkine(pu.ets(), q), oracle.dh_fkine(pu.dh, 0, q), atol=1e-13)


def test_oracle_matches_reference_run_fixtures():
    tool, base = tool_base()
    ch = chains.panda_ets(with_limits=True)
    q = REF["panda_q"]
    nt.assert_allclose(oracle.fkine(ch, q), REF["panda_fkine"], atol=1e-14)
    nt.assert_allclose(oracle.fkine(ch, q, base=base, tool=tool), REF["panda_fkine_bt"], atol=1e-14)
    nt.assert_allclose(oracle.jacob0(ch, q), REF["panda_jacob0"], atol=1e-14)
    nt.assert_allclose(oracle.jacobe(ch, q), REF["panda_jacobe"], atol=1e-14)
    nt.assert_allclose(oracle.jacob0(ch, q, tool=tool), REF["panda_jacob0_tool"], atol=1e-14)
    nt.assert_allclose(oracle.jacobe(ch, q, tool=tool), REF["panda_jacobe_tool"], atol=1e-14)
    nt.assert_allclose(oracle.hessian0(ch, q[:8]), REF["panda_hessian0"], atol=1e-14)
    mx = chains.Chain(mixed_spec())
    nt.assert_allclose(oracle.fkine(mx, REF["mixed_q"]), REF["mixed_fkine"], atol=1e-14)
    nt.assert_allclose(oracle.jacob0(mx, REF["mixed_q"]), REF["mixed_jacob0"], atol=1e-14)
    nt.assert_allclose(oracle.jacobe(mx, REF["mixed_q"]), REF["mixed_jacobe"], atol=1e-14)
    pu = chains.puma560()
    nt.assert_allclose(oracle.fkine(pu.ets(), REF["puma_q"]), REF["puma_fkine"], atol=1e-14)
    nt.assert_allclose(oracle.jacob0(pu.ets(), REF["puma_q"]), REF["puma_jacob0"], atol=1e-14)
    g = -pu.gravity
    args = (pu.L24(), 0, REF["puma_q"], REF["puma_qd"], REF["puma_qdd"])
    nt.assert_allclose(oracle.rne_dh(*args, g), REF["puma_rne"], rtol=1e-12, atol=1e-12)
    nt.assert_allclose(oracle.rne_dh(*args, g, [1, 2, 3, 1, 2, 3]), REF["puma_rne_fext"], rtol=1e-12, atol=1e-12)
    nt.assert_allclose(oracle.rne_dh(*args, np.zeros(3)), REF["puma_rne_g0"], rtol=1e-12, atol=1e-12)
    nt.assert_allclose(oracle.rne_dh(*args, -np.array([1.5, -2.0, -9.0])), REF["puma_rne_gx"], rtol=1e-12, atol=1e-12)
    pd = chains.panda_dh()
    nt.assert_allclose(oracle.fkine(pd.ets(), REF["pandadh_q"]), REF["pandadh_fkine"], atol=1e-14)
    args = (pd.L24(), 1, REF["pandadh_q"], REF["pandadh_qd"], REF["pandadh_qdd"])
    nt.assert_allclose(oracle.rne_dh(*args, -pd.gravity), REF["pandadh_rne"], rtol=1e-12, atol=1e-12)
    nt.assert_allclose(oracle.rne_dh(*args, -pd.gravity, [-1, 0.5, 2, 0.3, -0.2, 0.1]), REF["pandadh_rne_fext"],
                       rtol=1e-12, atol=1e-12)
    for name in ("rprp0", "rprp1", "prp0", "prp1"):
        mdh = int(name[-1])
        fext = [1, 2, 3, 4, 5, 6] if name.startswith("rprp") else None
        tau = oracle.rne_dh(REF[name + "_L24"], mdh, REF[name + "_q"], REF[name + "_qd"], REF[name + "_qdd"],
                            -np.array([0.5, -1.0, -9.81]), fext)
        nt.assert_allclose(tau, REF[name + "_rne"], rtol=1e-12, atol=1e-12)


def test_oracle_ik_matches_reference_run_fixtures():
    """Supplied q0 that converges in the first search: no RNG involved."""
    ch = chains.panda_ets(with_limits=True)
    dummy = np.zeros((101, 7))
    for meth, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):
        meta = REF["ik_%s_meta" % meth]
        hit = 0
        for i in range(len(meta)):
            if meta[i, 2] != 1 or meta[i, 0] != 1:
                continue
            q, sol, it, se, E = oracle.ik_lm(ch, REF["ik_Tep"][i], q0=REF["ik_q0"][i], k=k, method=meth, restarts=dummy)
            assert (sol, it, se) == tuple(meta[i])
            nt.assert_allclose(q, REF["ik_%s_q" % meth][i], atol=1e-8)
            hit += 1
        assert hit >= 15


def test_oracle_angle_axis_matches_reference_Angle_Axis():
    """fknm.Angle_Axis (fknm.cpp:112-162 -> ik.cpp:241-286) on random pairs, identical rotations (|li| = 0, tr > 0 -> zero),
    half turns about five axes (|li| ~ 0, tr <= 0 -> pi/2 (diag R + 1)) and rotations either side of the |li| < 1e-6 test."""
    e = np.array([oracle.angle_axis(a, b) for a, b in zip(PY["aa_Te"], PY["aa_Tep"])])
    assert np.all(np.abs(e - PY["aa_e"]).max(axis=1) <= angle_axis_tolerance(PY["aa_Te"], PY["aa_Tep"]))
    assert np.abs(e - PY["aa_e"])[PY["aa_tag"] <= 2].max() <= 5e-15
    tag = PY["aa_tag"]
    assert np.all(PY["aa_e"][tag == 1][:, 3:] == 0.0)                       # the (1,1,1) branch really was taken
    half = PY["aa_e"][tag == 2][:, 3:]
    assert np.all(np.abs(np.abs(half).max(axis=1) - np.pi) < 2.0) and np.all(half >= -1e-12)   # pi/2 (diag + 1) in [0, pi]
    assert np.any(np.all(PY["aa_e"][tag == 3][:, 3:] == 0.0, axis=1)) and np.any(np.abs(PY["aa_e"][tag == 3][:, 3:]).max(axis=1) > 3.0)


@pytest.mark.parametrize("key", sorted(PY_IK_CASES))
def test_oracle_python_ik_matches_reference_IK_py(key):
    """The NumPy restatement of the Python solvers (oracle.ikine_py; oracle.ikine_lm for the plain LM cases) against the
    reference's OWN robot/IK.py run under stand-in modules (oracle/ref_python.py) on explicit start tables: identical
    (success, iterations, searches) and q to 1e-7 -- across many searches, failures, joint-limit rejections, masks and the
    null-space terms kq / km / ps / pi.  Undamped NR / GN searches that start far from the solution are chaotic (1e-16
    differences decide whether a wandering search happens to land); there only what the first search decides is compared."""
    ch = chains.panda_ets(with_limits=True)
    prob, first, slimit, step, kw = PY_IK_CASES[key]
    Tep, tab = py_ik_problem(PY, key)
    meta, qref, Eref = PY[key + "_meta"], PY[key + "_q"], PY[key + "_E"]
    kw = dict(kw)
    we = kw.pop("mask", None)
    checked = 0
    for i in range(len(Tep)):
        o = oracle.ikine_py(ch, Tep[i], tab[i], step=step, slimit=slimit, we=we, **kw)
        if step not in ("lm", "qp") and not (meta[i, 0] == 1 and meta[i, 2] == 1):
            continue
        assert (o[1], o[2], o[3]) == tuple(meta[i]), (key, i)
        if meta[i, 0]:
            nt.assert_allclose(o[0], qref[i], atol=1e-7)
            assert abs(o[4] - Eref[i]) <= 1e-9 * max(1.0, abs(Eref[i]))
        checked += 1
        if step == "lm" and "kq" not in kw:                       # the C restatement of the same loop
            c = oracle.ikine_lm(ch, Tep[i], tab[i], slimit=slimit, we=we, ilimit=kw.get("ilimit", 30),
                                joint_limits=kw.get("joint_limits", True), k=kw["k"], method=kw["method"])
            assert (c[1], c[2], c[3]) == tuple(meta[i]), (key, i)
            if meta[i, 0]:
                nt.assert_allclose(c[0], qref[i], atol=1e-7)
    assert checked >= (len(Tep) if step in ("lm", "qp") else 6)


def test_oracle_qp_solver_against_scipy():
    """oracle/qp.py (the stand-in for qpsolvers/quadprog behind the reference's IK_QP.step: enumeration of active sets + KKT
    solves) against scipy.optimize SLSQP on random programmes of IK_QP's shape: same minimiser (to SLSQP's accuracy), same
    objective to 1e-10, constraints satisfied, and at least half of the programmes with an active row."""
    from scipy.optimize import minimize
    from oracle import qp
    rng = np.random.default_rng(5)
    active = 0
    for trial in range(30):
        n = 7
        J = rng.normal(size=(6, n)); e = rng.normal(size=6) * 0.3
        P = np.eye(n + 6); P[:n, :n] *= 0.01; P[n:, n:] = 1.0 / np.abs(e).sum() * np.eye(6)
        c = np.concatenate((rng.normal(size=n) * 0.01, np.zeros(6)))
        A = np.concatenate((J, np.eye(6)), axis=1)
        G = np.zeros((n + 6, n + 6)); h = np.zeros(n + 6)
        for i in range(n):
            if rng.random() < 0.5:
                G[i, i] = rng.choice([-1.0, 1.0]); h[i] = rng.uniform(-0.02, 0.05)
        x = qp.solve_qp(P, c, G, h, A, e)
        assert x is not None
        f = lambda z: 0.5 * z @ P @ z + c @ z
        res = minimize(f, np.zeros(n + 6), jac=lambda z: P @ z + c, method="SLSQP",
                       constraints=[{"type": "eq", "fun": lambda z: A @ z - e, "jac": lambda z: A},
                                    {"type": "ineq", "fun": lambda z: h - G @ z, "jac": lambda z: -G}], options={"ftol": 1e-15, "maxiter": 1000})
        assert np.abs(A @ x - e).max() < 1e-12 and np.all(G @ x - h <= 1e-12)
        assert f(x) <= f(res.x) + 1e-10 and np.abs(x - res.x).max() < 1e-5
        active += int(np.any(np.abs((G @ x - h)[np.any(G != 0, axis=1)]) < 1e-10))
    assert active >= 15


@pytest.mark.skipif(not ref_harness.available(), reason="oracle/_ref not built here")
def test_oracle_matches_live_reference_build():
    rng = np.random.default_rng(99)
    ch = chains.panda_ets()
    ref = ref_harness.RefETS(ch)
    q = rng.uniform(-np.pi, np.pi, (300, 7))
    nt.assert_allclose(oracle.fkine(ch, q), ref.fkine(q), atol=1e-14)
    nt.assert_allclose(oracle.jacob0(ch, q), ref.jacob0_batch(q), atol=1e-14)
    # single-vector shapes the reference accepts (test_ETS.py:359-362)
    for qq in (q[0], list(q[0]), q[0][None, :], q[0][:, None]):
        nt.assert_allclose(ref.jacob0(qq), oracle.jacob0(ch, q[0])[0], atol=1e-14)
        nt.assert_allclose(ref.fkine(qq), oracle.fkine(ch, q[0])[0], atol=1e-14)

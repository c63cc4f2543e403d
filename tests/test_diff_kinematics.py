"""jacob0_dot / manipulability / jacobm computed from the register-resident Jacobian.
Pins: reference tests/test_ERobot.py:28-51 (jacobm literal, ETS Panda), tests/test_ETS.py:4303-4334
(manipulability of the URDF Panda / Puma to 4 decimals), tests/test_DHRobot.py:1246-1262 (jacob0_dot ==
numerical Hessian . qd)."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import urdf
from oracle import oracle, chains
from helpers import literals, chain_from_ets, product_ets, mixed_spec

LIT = literals()


def _cases():
    panda = rtbhip.models.Panda().ets()
    puma = rtbhip.models.DH.Puma560().ets()
    ur5 = urdf.load("UR5").ets()
    fetch = urdf.load("Fetch").ets()                       # 10 joints: the one-wave-per-SIMD instantiations
    return [("panda", panda, chains.panda_ets()), ("puma", puma, chain_from_ets(puma)), ("ur5", ur5, chain_from_ets(ur5)),
            ("fetch", fetch, chain_from_ets(fetch))]


def test_oracle_pins():
    ch = chains.panda_ets()
    q1 = LIT["K_panda_jacobm_q"]
    nt.assert_array_almost_equal(oracle.jacobm(ch, q1)[0].reshape(7, 1), LIT["K_panda_jacobm"])
    # jacob0_dot == (numerical Hessian of jacob0) . qd  (reference tests/test_jacob.py:73-81)
    qd = np.array([0.1, -0.2, 0.3, -0.4, 0.5, -0.6, 0.7])
    h = 1e-6
    Jd = np.zeros((6, 7))
    for i in range(7):
        d = np.zeros(7); d[i] = h
        Jd += (oracle.jacob0(ch, q1 + d)[0] - oracle.jacob0(ch, q1 - d)[0]) / (2 * h) * qd[i]
    nt.assert_array_almost_equal(oracle.jacob_dot(ch, q1, qd)[0], Jd, decimal=6)


def test_manipulability_goldens_reference_models():
    """reference tests/test_ETS.py:4303-4353: URDF Panda at qr, URDF Puma560 at qn."""
    import emu_harness as emu
    p = urdf.load("Panda")
    qr = np.array([0, -0.3, 0, -2.2, 0, 2.0, np.pi / 4])
    nt.assert_almost_equal(emu.diff(p.ets(), 1, qr, axes=63 | (2 << 8))[0], 0.11222, decimal=4)      # test_cond
    nt.assert_almost_equal(emu.diff(p.ets(), 1, qr, axes=63 | (1 << 8))[0], 0.209013, decimal=4)     # test_minsingular
    for axes, mask, want in (("all", 63, 0.0837), ("trans", 7, 0.1438), ("rot", 56, 2.7455)):
        nt.assert_almost_equal(emu.diff(p.ets(), 1, qr, axes=mask)[0], want, decimal=4)
        nt.assert_almost_equal(oracle.manipulability(chain_from_ets(p.ets()), qr, axes)[0], want, decimal=4)
    pu = urdf.load("Puma560")
    qn = np.array([0, np.pi / 4, np.pi, 0, np.pi / 4, 0])
    for mask, want in ((63, 0.0805), (7, 0.1354), (56, 2.44949)):
        nt.assert_almost_equal(emu.diff(pu.ets(), 1, qn, axes=mask)[0], want, decimal=4)


@pytest.mark.parametrize("name", ["panda", "puma", "ur5", "mixed", "fetch10", "gen3_9"])
def test_emu_vs_oracle(name):
    import emu_harness as emu
    if name == "mixed":
        spec = mixed_spec()
        e, ch = product_ets(spec), chains.Chain(spec)
    elif name in ("fetch10", "gen3_9"):
        e = urdf.load("Fetch" if name == "fetch10" else "KinovaGen3").ets()
        ch = chain_from_ets(e)
    else:
        _, e, ch = [c for c in _cases() if c[0] == name][0]
    rng = np.random.default_rng(3)
    q, qd = rng.uniform(-2, 2, (20, e.n)), rng.normal(size=(20, e.n))
    for frame in (0, 1):
        nt.assert_allclose(emu.diff(e, 0, q, qd, frame=frame), oracle.jacob_dot(ch, q, qd, frame=frame), atol=1e-11)
    for axes, mask in (("all", 63), ("trans", 7), ("rot", 56), ([True, False, True, True, False, True], 45)):
        m, ref = emu.diff(e, 1, q, axes=mask), oracle.manipulability(ch, q, axes)
        nt.assert_allclose(m, ref, rtol=1e-9, atol=1e-12)
    for axes, mask in (("all", 63), ("trans", 7), ("rot", 56), ([True, False, True, True, False, True], 45)):
        for method, code in (("minsingular", 1), ("invcondition", 2)):
            m, ref = emu.diff(e, 1, q, axes=mask | (code << 8)), oracle.manipulability(ch, q, axes, method=method)
            nt.assert_allclose(m, ref, rtol=1e-7, atol=1e-9)
    if e.n >= 6:
        for axes, mask in (("all", 63), ("trans", 7), ("rot", 56)):
            jm, ref = emu.diff(e, 2, q, axes=mask), oracle.jacobm(ch, q, axes)
            nt.assert_allclose(jm, ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max()))
    nt.assert_array_almost_equal(emu.diff(rtbhip.models.Panda().ets(), 2, LIT["K_panda_jacobm_q"])[0].reshape(7, 1),
                                 LIT["K_panda_jacobm"])


def _central_rate(ch, q, qd, rep, h=1e-5):
    """An accurate derivative of the analytical Jacobian along qd (central differences of the oracle), to show that the
    forward-difference quantity the reference defines is what it claims to be (reference test: 4 decimals, tests/test_jacob.py:83-97)."""
    Jd = np.zeros((6, ch.n))
    for i in range(ch.n):
        d = np.zeros(ch.n); d[i] = h
        Jd += (oracle.jacob0_analytical(ch, q + d, rep)[0] - oracle.jacob0_analytical(ch, q - d, rep)[0]) / (2 * h) * qd[i]
    return Jd


@pytest.mark.parametrize("rep", ["rpy/xyz", "rpy/zyx", "eul", "exp"])
def test_emu_jacob0_dot_with_representation(rep):
    """Robot.jacob0_dot(q, qd, representation=...) (robot/Robot.py:1065-1098): forward-difference numerical Hessian of
    jacob0_analytical contracted with qd.  The device function against the NumPy restatement of exactly that (both carry the
    1e-8 step's round-off, so they agree to ~1e-6), and against an accurate derivative at the reference test's 4 decimals;
    ETS Puma560 at the reference's own q / qd (tests/test_jacob.py:19-22), plus the Panda."""
    import emu_harness as emu
    code = {"rpy/xyz": 0, "rpy/zyx": 1, "eul": 2, "exp": 3}[rep]
    puma = rtbhip.models.DH.Puma560().ets()
    for e, ch, q, qd in ((puma, chain_from_ets(puma), np.array([0.1, 0.2, 0.3, 0.1, 0.2, 0.3]), np.array([0.1, -0.2, 0.3, -0.1, 0.2, -0.3])),
                         (rtbhip.models.Panda().ets(), chains.panda_ets(), LIT["K_panda_jacobm_q"], np.array([0.1, -0.2, 0.3, -0.4, 0.5, -0.6, 0.7]))):
        Jd = emu.diff(e, 4, q, qd=qd, axes=code)[0]
        nt.assert_allclose(Jd, oracle.jacob0_dot_analytical(ch, q, qd, rep)[0], atol=5e-6)
        nt.assert_array_almost_equal(Jd, _central_rate(ch, q, qd, rep), decimal=4)
        # translational rows of the analytical Jacobian are those of jacob0: their rate is the geometric one
        nt.assert_allclose(Jd[:3], oracle.jacob_dot(ch, q, qd)[0][:3], atol=5e-6)


@pytest.mark.gpu
def test_gpu_jacob0_dot_representation_and_hessian_from_jacobian():
    import torch
    rng = np.random.default_rng(12)
    for name, e, ch in _cases():
        N = 130
        q, qd = rng.uniform(-1.2, 1.2, (N, e.n)), rng.normal(size=(N, e.n))
        for rep in ("rpy/xyz", "rpy/zyx", "eul", "exp"):
            Jd = e.jacob0_dot(q, qd, representation=rep)
            assert Jd.shape == (N, 6, e.n)
            ref = oracle.jacob0_dot_analytical(ch, q[:40], qd[:40], rep)
            scale = max(1.0, np.abs(ref).max())
            nt.assert_allclose(Jd[:40], ref, atol=2e-5 * scale)
            nt.assert_array_equal(e.jacob0_dot(torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda(), representation=rep).cpu().numpy(), Jd)
        one = e.jacob0_dot(q[0], qd[0], representation="eul")
        assert one.shape == (6, e.n)
        with pytest.raises(ValueError):
            e.jacob0_dot(q[0], qd[0], representation="quaternion")
        # the reference's calling form hessian0(q, J0=J0) / hessiane(Je=...) (robot/Robot.py:1069, robot/ETS.py:1419,1537)
        J0, Je = e.jacob0(q), e.jacobe(q)
        nt.assert_array_equal(e.hessian0(J0=J0), e.hessian0(q))
        nt.assert_array_equal(e.hessiane(Je=Je), e.hessiane(q))
        nt.assert_allclose(e.hessian0(J0=J0)[:30], oracle.hessian(ch, q[:30]), atol=1e-10)
        H1 = e.hessian0(J0=J0[0])
        assert H1.shape == (e.n, 6, e.n)
        Ht = rtbhip.hessian_from_jacobian(torch.from_numpy(J0).cuda())
        nt.assert_array_equal(Ht.cpu().numpy(), e.hessian0(q))
        with pytest.raises(ValueError):
            e.hessian0()
        with pytest.raises(ValueError):
            e.hessian0(J0=np.zeros((6, e.n + 1)))
    # 14 joints: the run-time-n lane loop
    J14 = rng.normal(size=(70, 6, 14))
    H14 = rtbhip.hessian_from_jacobian(J14)
    import emu_harness as emu
    nt.assert_allclose(H14, emu.hess_from_jac(J14), atol=1e-13)
    # p_servo, method "angle-axis" (tools/p_servo.py:46-117)
    Te, Tep = e.eval(q[:50]), e.eval(q[:50] + 0.01)
    v, arrived = rtbhip.p_servo(Te, Tep, gain=2.0, threshold=0.5, method="angle-axis")
    nt.assert_allclose(v, 2.0 * rtbhip.angle_axis(Te, Tep), atol=0)
    assert arrived.shape == (50,) and arrived.dtype == bool
    v1, a1 = rtbhip.p_servo(Te[0], Tep[0], gain=[1, 1, 1, 2, 2, 2], threshold=0.5, method="angle-axis")
    assert v1.shape == (6,) and isinstance(a1, bool)
    # (the reference's default method "rpy": tests/test_p_servo.py)


@pytest.mark.gpu
def test_gpu_goldens_shapes_errors():
    panda = rtbhip.models.Panda()
    q1 = LIT["K_panda_jacobm_q"]
    for q in (q1, list(q1), q1.reshape(1, 7), q1.reshape(7, 1)):            # reference tests/test_ERobot.py:48-51
        jm = panda.jacobm(q)
        assert jm.shape == (7, 1)
        nt.assert_array_almost_equal(jm, LIT["K_panda_jacobm"])
    p = urdf.load("Panda")
    qr = np.array([0, -0.3, 0, -2.2, 0, 2.0, np.pi / 4])
    nt.assert_almost_equal(p.ets().manipulability(qr), 0.0837, decimal=4)
    nt.assert_almost_equal(p.ets().manipulability(qr, axes="trans"), 0.1438, decimal=4)
    nt.assert_almost_equal(p.ets().manipulability(qr, axes="rot"), 2.7455, decimal=4)
    m2 = p.ets().manipulability(np.c_[qr, qr].T)
    assert m2.shape == (2,)
    with pytest.raises(ValueError):
        panda.manipulability(qr, axes="abcdef")
    with pytest.raises(ValueError):
        panda.manipulability(qr, method="nonsense")
    # reference tests/test_ETS.py:4339-4353 (URDF Panda at qr)
    nt.assert_almost_equal(p.ets().manipulability(qr, method="invcondition"), 0.11222, decimal=4)
    nt.assert_almost_equal(p.ets().manipulability(qr, method="minsingular"), 0.209013, decimal=4)
    seventeen = rtbhip.DHRobot([rtbhip.RevoluteDH(a=0.1) for _ in range(17)]).ets()
    from helpers import large_sizes_served
    if large_sizes_served():                   # beyond the built-in sizes: instantiated at run time (tests/test_large_chains_gpu.py)
        assert np.isfinite(seventeen.manipulability(np.linspace(0.1, 1.0, 17)))
    else:
        with pytest.raises(rtbhip.RtbHipError):
            seventeen.manipulability(np.zeros(17))
    # 11..16 joints: the spilling instantiations of the differential-kinematics consumers
    rng = np.random.default_rng(3)
    arm14 = rtbhip.DHRobot([rtbhip.RevoluteDH(a=0.05 + 0.01 * k, d=0.05, alpha=[0.0, np.pi / 2, -np.pi / 2][k % 3]) for k in range(14)]).ets()
    ch14 = chain_from_ets(arm14)
    q, qd = rng.uniform(-1.5, 1.5, (70, 14)), rng.normal(size=(70, 14))
    nt.assert_allclose(arm14.jacob0_dot(q, qd)[:10], oracle.jacob_dot(ch14, q[:10], qd[:10]), atol=1e-10)
    nt.assert_allclose(arm14.manipulability(q)[:10], oracle.manipulability(ch14, q[:10]), rtol=1e-8, atol=1e-12)
    ref = oracle.jacobm(ch14, q[:10])
    nt.assert_allclose(arm14.jacobm(q)[:10], ref, rtol=1e-6, atol=1e-8 * max(1.0, np.abs(ref).max()))
    nt.assert_allclose(arm14.jacob0_analytical(q, "eul")[:10], oracle.jacob0_analytical(ch14, q[:10], "eul"), atol=1e-9)
    nt.assert_allclose(arm14.hessian0(q)[:5], oracle.hessian(ch14, q[:5]), atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 65, 4097])
def test_gpu_vs_oracle(N):
    import torch
    for name, e, ch in _cases():
        rng = np.random.default_rng(N)
        q, qd = rng.uniform(-2, 2, (N, e.n)), rng.normal(size=(N, e.n))
        k = min(N, 300)
        Jd = e.jacob0_dot(q, qd).reshape(N, 6, e.n)
        nt.assert_allclose(Jd[:k], oracle.jacob_dot(ch, q[:k], qd[:k]), atol=1e-10)
        nt.assert_allclose(e.jacobe_dot(q, qd).reshape(N, 6, e.n)[:k], oracle.jacob_dot(ch, q[:k], qd[:k], frame=1), atol=1e-10)
        # J is linear in qd and equals H . qd of the batched Hessian kernel
        H = e.hessian0(q).reshape(N, e.n, 6, e.n)
        nt.assert_allclose(Jd, np.einsum("njrc,nj->nrc", H, qd), atol=1e-11)
        for axes in ("all", "trans", "rot"):
            m = np.atleast_1d(e.manipulability(q, axes=axes))
            nt.assert_allclose(m[:k], oracle.manipulability(ch, q[:k], axes), rtol=1e-9, atol=1e-12)
            for method in ("minsingular", "invcondition"):
                m = np.atleast_1d(e.manipulability(q, axes=axes, method=method))
                nt.assert_allclose(m[:k], oracle.manipulability(ch, q[:k], axes, method=method), rtol=1e-7, atol=1e-9)
            jm = e.jacobm(q, axes=axes).reshape(N, e.n)
            ref = oracle.jacobm(ch, q[:k], axes)
            nt.assert_allclose(jm[:k], ref, rtol=1e-7, atol=1e-9 * max(1.0, np.abs(ref).max()))
        # jacobm is the gradient of the manipulability (central differences on a few rows)
        h = 1e-6
        for i in range(min(N, 3)):
            g = np.array([(e.manipulability(q[i] + h * np.eye(e.n)[c]) - e.manipulability(q[i] - h * np.eye(e.n)[c])) / (2 * h)
                          for c in range(e.n)])
            nt.assert_allclose(e.jacobm(q[i]).reshape(-1), g, atol=2e-6 * max(1.0, np.abs(g).max()))
        qt, qdt = torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()
        nt.assert_array_equal(e.jacob0_dot(qt, qdt).cpu().numpy().reshape(N, 6, e.n), Jd)
        nt.assert_array_equal(np.atleast_1d(e.manipulability(qt).cpu().numpy() if N > 1 else e.manipulability(qt)),
                              np.atleast_1d(e.manipulability(q)))


@pytest.mark.gpu
def test_gpu_dhrobot_passthroughs_puma_goldens():
    """reference tests/test_DHRobot.py:1264-1290 (test_yoshi on the DH Puma560 at qn) and an IK round trip."""
    puma = rtbhip.models.DH.Puma560()
    qn = puma.qn
    nt.assert_almost_equal(puma.manipulability(qn), 0.0786, decimal=4)
    m2 = puma.manipulability(np.c_[qn, qn].T)
    nt.assert_almost_equal(m2[0], 0.0786, decimal=4)
    nt.assert_almost_equal(puma.manipulability(qn, axes="trans"), 0.111181, decimal=4)
    nt.assert_almost_equal(puma.manipulability(qn, axes="rot"), 2.44949, decimal=4)
    Tep = puma.fkine(qn)
    for fn in (puma.ik_LM, puma.ik_GN, puma.ik_NR):
        q, ok, it, se, E = fn(Tep, seed=1)
        assert ok == 1 and E < 1e-6
        nt.assert_allclose(puma.fkine(q), Tep, atol=2e-3)
    sol = puma.ikine_LM(Tep, seed=1)
    assert sol.success and sol.residual < 1e-6
    qd = [0.1, -0.2, 0.3, -0.4, 0.5, -0.6]
    H = puma.hessian0(qn)
    nt.assert_allclose(puma.jacob0_dot(qn, qd), np.tensordot(H, qd, (0, 0)), atol=1e-12)


# ---------------------------------------------------------------- partial_fkine0 (ETS.py:1821-2013)
def test_oracle_partial_fkine3_golden():
    """tests/test_ETS.py:1796-4263: the (7,7,6,7) literal for the Panda at q1 (assert_almost_equal, 7 dp)."""
    q = np.array(LIT["panda_q"])
    ans = np.array(LIT["K_panda_partial_fkine3"])
    assert ans.shape == (7, 7, 6, 7)
    got = oracle.partial_fkine0(chains.panda_ets(), q, 3)
    nt.assert_almost_equal(got, ans)
    # orders 1 and 2 are the Jacobian and the Hessian
    nt.assert_allclose(oracle.partial_fkine0(chains.panda_ets(), q, 2), oracle.hessian0(chains.panda_ets(), q)[0])


def test_emu_partial_fkine_matches_oracle_and_golden():
    import emu_harness as emu
    ch = chains.panda_ets()
    ets = rtbhip.models.Panda().ets()
    q = np.array(LIT["panda_q"])
    nt.assert_almost_equal(emu.partial(ets, q, 3)[0], np.array(LIT["K_panda_partial_fkine3"]))
    rng = np.random.default_rng(5)
    qs = rng.uniform(-np.pi, np.pi, (3, 7))
    tool = chains.elementary("tx", 0.1) @ chains.elementary("Ry", 0.3)
    got = emu.partial(ets, qs, 3, tool=tool)
    for i in range(3):
        nt.assert_allclose(got[i], oracle.partial_fkine0(ch, qs[i], 3, tool=tool), atol=1e-12)
    # order 4 on a short chain with every joint kind, against the oracle and a finite difference of order 3
    spec = [("Rz", None), ("tx", 0.3), ("Ry", None, True), ("tz", None), ("Rx", None), ("ty", 0.2)]
    short = chains.Chain(spec, name="short4")
    es = product_ets(spec)
    q4 = rng.uniform(-1, 1, (2, 4))
    g4 = emu.partial(es, q4, 4)
    assert g4.shape == (2, 4, 4, 4, 6, 4)
    for i in range(2):
        nt.assert_allclose(g4[i], oracle.partial_fkine0(short, q4[i], 4), atol=1e-12)
    g5 = emu.partial(es, q4, 5)
    nt.assert_allclose(g5[1], oracle.partial_fkine0(short, q4[1], 5), atol=1e-12)


@pytest.mark.gpu
def test_gpu_partial_fkine0():
    import torch
    panda = rtbhip.models.Panda()
    q1 = np.array(LIT["panda_q"])
    ans = np.array(LIT["K_panda_partial_fkine3"])
    nt.assert_almost_equal(panda.ets().partial_fkine0(q1, 3), ans)            # reference tests/test_ETS.py:4259-4263
    nt.assert_almost_equal(panda.partial_fkine0(q1, 3), ans)                  # Robot level (r2 in the same test)
    nt.assert_array_equal(panda.ets().partial_fkine0(q1, 1), panda.ets().jacob0(q1))
    nt.assert_array_equal(panda.ets().partial_fkine0(q1, 2), panda.ets().hessian0(q1))
    rng = np.random.default_rng(11)
    for name, e, ch in _cases()[:3]:
        N = 130
        q = rng.uniform(-2, 2, (N, e.n))
        P3 = e.partial_fkine0(q, 3)
        assert P3.shape == (N, e.n, e.n, 6, e.n)
        for i in (0, 63, 64, 129):
            nt.assert_allclose(P3[i], oracle.partial_fkine0(ch, q[i], 3), atol=1e-11)
        # (the reference's tensor is the product rule applied to H[k,:,j] = J_w[:,k] x J[:,j] for every index order; it is
        # the true derivative of its Hessian only where that expression is, so there is no finite-difference property to test)
        qt = torch.from_numpy(q).cuda()
        nt.assert_array_equal(e.partial_fkine0(qt, 3).cpu().numpy(), P3)
    e, ch = _cases()[2][1:]                                                   # UR5: orders 4 and 5
    q = rng.uniform(-2, 2, (3, e.n))
    P4 = e.partial_fkine0(q, 4)
    nt.assert_allclose(P4[2], oracle.partial_fkine0(ch, q[2], 4), atol=1e-11)
    P5 = e.partial_fkine0(q[:2], 5)
    assert P5.shape == (2,) + (6,) * 4 + (6, 6)
    nt.assert_allclose(P5[0], oracle.partial_fkine0(ch, q[0], 5), atol=1e-11)
    with pytest.raises(rtbhip.RtbHipError):
        e.partial_fkine0(q, 7)
    assert e.partial_fkine0(np.zeros((0, 6)), 3).shape == (0, 6, 6, 6, 6)


# ---------------------------------------------------------------- manipulability(J=) / jacobm(J=, H=): from the caller's arrays
def _reference_manipulability(J, method, axes):
    """robot/Robot.py:848-869 as it stands: yoshikawa / condition / minsingular on J[axes, :]"""
    Ja = J[np.asarray(axes, dtype=bool), :]
    if method == "yoshikawa":
        return abs(np.linalg.det(Ja)) if Ja.shape[0] == Ja.shape[1] else np.sqrt(abs(np.linalg.det(Ja @ Ja.T)))
    if method == "invcondition":
        return 1 / np.linalg.cond(Ja)
    return np.linalg.svd(Ja, compute_uv=False)[-1]


def _reference_jacobm(J, H, axes):
    """robot/Robot.py:1213-1233 as it stands (H is the (n,6,n) tensor hessian0 returns)"""
    axes = np.asarray(axes, dtype=bool)
    m = _reference_manipulability(J, "yoshikawa", axes)
    Ja, Ha = J[axes, :], H[:, axes, :]
    b = np.linalg.inv(Ja @ Ja.T)
    return np.array([m * (Ja @ Ha[i].T).flatten("F") @ b.flatten("F") for i in range(J.shape[1])])


_AXES = {"all": [1] * 6, "trans": [1, 1, 1, 0, 0, 0], "rot": [0, 0, 0, 1, 1, 1]}


def _from_jacobian_checks(ets, ch, N, seed, dev=None):
    """manipulability_from_jacobian / jacobm_from_jacobian against the reference's NumPy expressions on Jacobians of random configurations;
    `dev` turns host arrays into device tensors."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-2.5, 2.5, (N, ch.n))
    J = oracle.jacob0(ch, q)
    H = oracle.hessian0(ch, q)
    put = (lambda a: a) if dev is None else dev
    get = (lambda a: a) if dev is None else (lambda t: t.cpu().numpy())
    for method in ("yoshikawa", "minsingular", "invcondition"):
        for axes in ("all", "trans", "rot", [True, False, True, True, False, True]):
            ax = _AXES.get(axes, axes) if isinstance(axes, str) else axes
            if ch.n < sum(bool(a) for a in ax) and method != "yoshikawa":
                pass                                                   # more rows than joints: the singular values are those of J_a^T J_a, still compared
            got = get(rtbhip.manipulability_from_jacobian(put(J), method=method, axes=axes))
            want = np.array([_reference_manipulability(J[i], method, ax) for i in range(N)])
            nt.assert_allclose(got, want, rtol=1e-8, atol=1e-10)
    one = rtbhip.manipulability_from_jacobian(J[0])
    assert isinstance(one, float) and abs(one - _reference_manipulability(J[0], "yoshikawa", [1] * 6)) < 1e-12
    for axes in ("all", "trans", "rot"):
        ax = _AXES[axes]
        if ch.n < sum(ax):
            continue                                                   # J_a J_a^T is singular with fewer joints than rows: the reference's inv() fails too
        want = np.array([_reference_jacobm(J[i], H[i], ax) for i in range(N)])
        ok = np.array([np.linalg.cond(J[i][np.asarray(ax, bool)] @ J[i][np.asarray(ax, bool)].T) < 1e8 for i in range(N)])
        scale = np.abs(want[ok]).max()
        nt.assert_allclose(get(rtbhip.jacobm_from_jacobian(put(J), axes=axes))[ok], want[ok], atol=1e-8 * scale)            # H formed from J
        nt.assert_allclose(get(rtbhip.jacobm_from_jacobian(put(J), H=put(H), axes=axes))[ok], want[ok], atol=1e-8 * scale)   # H supplied
        # a Hessian that is NOT the Jacobian's own is honoured (the reference contracts whatever it is given)
        H2 = H * 0.5
        nt.assert_allclose(get(rtbhip.jacobm_from_jacobian(put(J), H=put(H2), axes=axes))[ok], 0.5 * want[ok], atol=1e-8 * scale)
    col = rtbhip.jacobm_from_jacobian(J[0], H=H[0])
    assert col.shape == (ch.n, 1)


@pytest.mark.parametrize("name", ["panda", "puma", "fetch"])
def test_emu_manipulability_and_jacobm_from_a_supplied_jacobian(name):
    """the kernel body of k_diff_from_jac replayed on the CPU (tests/cpu_backend.py) against the reference's NumPy expressions"""
    import cpu_backend
    _, ets, ch = [c for c in _cases() if c[0] == name][0]
    with cpu_backend.installed() as be:
        _from_jacobian_checks(ets, ch, 130, 11)
        assert be.calls.get("rtbhip_manipulability_from_jacobian", 0) > 0 and be.calls.get("rtbhip_jacobm_from_jacobian", 0) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["panda", "puma", "ur5", "fetch"])
def test_gpu_manipulability_and_jacobm_from_a_supplied_jacobian(name):
    """Robot.manipulability(J=) / Robot.jacobm(J=, H=) (robot/Robot.py:896, :1201-1233) on the device, host arrays and device tensors"""
    import torch
    from helpers import DEV
    _, ets, ch = [c for c in _cases() if c[0] == name][0]
    _from_jacobian_checks(ets, ch, 1000, 12)
    _from_jacobian_checks(ets, ch, 129, 13, dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV()).cuda())


@pytest.mark.gpu
def test_gpu_robot_level_J_and_H_keywords_and_errors():
    """the pass-throughs: robot.manipulability(J=), robot.jacobm(J=, H=), robot.jacobm(q, H=) for an ETS robot, a DH robot and a URDF robot"""
    rng = np.random.default_rng(3)
    for robot in (rtbhip.models.Panda(), rtbhip.models.DH.Puma560(), urdf.load("UR5")):
        n = robot.n if not isinstance(robot, urdf.URDFRobot) else robot.ets().n
        q = rng.uniform(-2, 2, n)
        J, H = np.asarray(robot.jacob0(q)), np.asarray(robot.hessian0(q))
        for axes in ("all", "trans", "rot"):
            nt.assert_allclose(robot.manipulability(J=J, axes=axes), _reference_manipulability(J, "yoshikawa", _AXES[axes]), rtol=1e-9)
            want = _reference_jacobm(J, H, _AXES[axes])
            nt.assert_allclose(np.asarray(robot.jacobm(J=J, axes=axes)).ravel(), want, atol=1e-9 * np.abs(want).max())
            nt.assert_allclose(np.asarray(robot.jacobm(J=J, H=H, axes=axes)).ravel(), want, atol=1e-9 * np.abs(want).max())
        nt.assert_allclose(robot.manipulability(J=J, method="minsingular"), np.linalg.svd(J, compute_uv=False)[-1], rtol=1e-8)
    with pytest.raises(ValueError):
        rtbhip.manipulability_from_jacobian(np.zeros((5, 7)))
    with pytest.raises(ValueError):
        rtbhip.manipulability_from_jacobian(np.zeros((6, 7)), method="asada")
    with pytest.raises(ValueError):
        rtbhip.jacobm_from_jacobian(np.zeros((6, 7)), H=np.zeros((6, 7, 7)))
    with pytest.raises(rtbhip.RtbHipError):
        rtbhip.manipulability_from_jacobian(np.zeros((6, 17)))
    with pytest.raises(rtbhip.RtbHipError):
        rtbhip.manipulability_from_jacobian(np.zeros((6, 7)), axes=[False] * 6)
    assert rtbhip.manipulability_from_jacobian(np.zeros((0, 6, 7))).shape == (0,)


def test_emu_inverse_condition_of_a_block_that_is_zero_but_for_rounding():
    """A single Rx joint behind two general constants, with a tool whose translation lies exactly along the joint axis: the translational Jacobian is
    a zero vector.  The reference's column formula multiplies exact zeros of the tool (cond = inf, 1 / cond = 0: ETS.py:1789-1791); the one-walk form
    z x (p_e - p) leaves 1e-17 of rounding, whose "condition number" would be 1.  Found by scripts/fuzz_more.py (profiles/r06_aj_fuzz_more.txt)."""
    import emu_harness as emu
    A = chains.elementary("Rz", 0.7) @ chains.elementary("Ry", -0.4) @ chains.elementary("tx", 0.15) @ chains.elementary("tz", -0.1)
    B = chains.elementary("Rx", 1.1) @ chains.elementary("Rz", -0.9) @ chains.elementary("ty", 0.09)
    spec = [A, B, ("Rx", None, False)]
    e, ch = product_ets(spec), chains.Chain(spec)
    tool = chains.elementary("tx", 0.17) @ chains.elementary("Rx", 0.6)
    q = np.random.default_rng(5).uniform(-2.5, 2.5, (12, 1))
    ref = oracle.manipulability(ch, q, "trans", tool=tool, method="invcondition")
    nt.assert_array_equal(ref, 0.0)
    nt.assert_array_equal(emu.diff(e, 1, q, axes=7 | (2 << 8), tool=tool), 0.0)
    nt.assert_array_equal(emu.diff(e, 1, q, axes=7 | (1 << 8), tool=tool) < 1e-15, True)          # the smallest singular value: rounding level
    # and a block that is NOT zero keeps its value
    nt.assert_allclose(emu.diff(e, 1, q, axes=56 | (2 << 8), tool=tool), 1.0, rtol=1e-12)

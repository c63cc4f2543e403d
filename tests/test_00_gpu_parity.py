"""-m gpu: the parity tests proper.  Everything goes through the C ABI (ctypes -> librtbhip.so ->
gfx950 kernels) and is compared with the CPU oracle on the same seeded inputs, with the committed
golden fixtures, and -- at BASELINE.json's full sizes -- through size-independent properties.

Tolerances (fp64; north_star: <= 1e-10 max pose error vs the reference CPU path):
    T, J, H : max abs error <= 1e-10  (observed ~1e-15)
    tau     : <= 1e-9 relative to max |tau|
"""
import numpy as np
import numpy.testing as nt
import pytest
import torch

import rtbhip
from rtbhip import urdf
from oracle import oracle, chains
from helpers import literals, ref_outputs, mixed_spec, product_ets, tool_base, DEV, full_size

pytestmark = pytest.mark.gpu
TOL = 1e-10
LIT = literals()
REF = ref_outputs()


def test_native_library_is_the_path():
    assert rtbhip.device_count() >= 1
    import ctypes
    # the loaded shared object is the in-tree HIP library, not a fallback
    assert rtbhip.lib()._name.endswith("robotics-toolbox-python_amd/lib/librtbhip.so")
    g, b, l = rtbhip.last_launch()
    p = rtbhip.models.Panda()
    p.fkine(np.zeros(7))
    g, b, l = rtbhip.last_launch()
    assert (g, b) == (1, 64) and l > 0
    assert rtbhip.lib().rtbhip_version() >= 100


def test_golden_literals_G1_G2_G3_G8():
    p = rtbhip.models.Panda()
    q1 = LIT["panda_q"]
    for q in (q1, list(q1), q1[None, :], q1[:, None]):                     # test_ETS.py:359-362
        nt.assert_array_almost_equal(p.fkine(q), LIT["G1_panda_fkine"], decimal=6)
        nt.assert_array_almost_equal(p.jacob0(q), LIT["G2_panda_jacob0"], decimal=6)
        assert p.jacob0(q).shape == (6, 7) and p.fkine(q).shape == (4, 4)
    with pytest.raises(TypeError):
        p.jacob0("Wfgsrth")                                                   # test_ETS.py:363
    T = p.fkine(q1)
    tr2jac = np.zeros((6, 6)); tr2jac[:3, :3] = T[:3, :3].T; tr2jac[3:, 3:] = T[:3, :3].T
    nt.assert_array_almost_equal(p.jacobe(q1), tr2jac @ p.jacob0(q1), decimal=12)   # test_ETS.py:365-398
    raw = LIT["G8_panda_hessian0_raw"]
    nt.assert_array_almost_equal(p.hessian0(q1), np.stack([raw[:, :, i] for i in range(7)]), decimal=6)
    # hessian with the ee segment as `tool` (test_ETS.py:1130-1579)
    arm = rtbhip.ETS(p.ets()[:-2])
    ee = chains.elementary("tz", 103 * 1e-3) @ chains.elementary("Rz", -np.pi / 4)
    raw = LIT["G8_panda_hessian0_tool_raw"]
    nt.assert_array_almost_equal(arm.hessian0(q1, tool=ee), np.stack([raw[:, :, i] for i in range(7)]), decimal=6)


def test_reference_run_fixtures_panda():
    tool, base = tool_base()
    ets = rtbhip.models.Panda().ets()
    q = REF["panda_q"]
    nt.assert_allclose(ets.eval(q), REF["panda_fkine"], atol=TOL)
    nt.assert_allclose(ets.eval(q, base=base, tool=tool), REF["panda_fkine_bt"], atol=TOL)
    nt.assert_allclose(ets.jacob0(q), REF["panda_jacob0"], atol=TOL)
    nt.assert_allclose(ets.jacobe(q), REF["panda_jacobe"], atol=TOL)
    nt.assert_allclose(ets.jacob0(q, tool=tool), REF["panda_jacob0_tool"], atol=TOL)
    nt.assert_allclose(ets.jacobe(q, tool=tool), REF["panda_jacobe_tool"], atol=TOL)
    nt.assert_allclose(ets.hessian0(q[:8]), REF["panda_hessian0"], atol=TOL)
    T, J = ets.fkine_jacob0(q, base=base, tool=tool)
    nt.assert_allclose(T, REF["panda_fkine_bt"], atol=TOL)       # base reaches T ...
    nt.assert_allclose(J, REF["panda_jacob0_tool"], atol=TOL)    # ... but never the Jacobian


def test_mixed_chain_all_axes_flips_se3():
    ets = product_ets(mixed_spec())
    q = REF["mixed_q"]
    nt.assert_allclose(ets.eval(q), REF["mixed_fkine"], atol=TOL)
    nt.assert_allclose(ets.jacob0(q), REF["mixed_jacob0"], atol=TOL)
    nt.assert_allclose(ets.jacobe(q), REF["mixed_jacobe"], atol=TOL)


def test_batch_equals_row_by_row():
    """test_ETS.py:236-260: 6-joint Rx,Ry,Rz,tx,ty,tz robot, qt = arange(60).reshape(10,6)."""
    ET = rtbhip.ET
    r = ET.Rx() * ET.Ry() * ET.Rz() * ET.tx() * ET.ty() * ET.tz()
    qt = np.arange(60.0).reshape(10, 6)
    TT = r.eval(qt)
    JJ = r.jacob0(qt)
    for i in range(10):
        nt.assert_allclose(TT[i], r.eval(qt[i]), atol=1e-14)
        nt.assert_allclose(JJ[i], r.jacob0(qt[i]), atol=1e-14)


@pytest.mark.parametrize("N", [1, 63, 64, 65, 1000, 4097])
def test_ragged_sizes_vs_oracle(N):
    ets = rtbhip.models.Panda().ets()
    ch = chains.panda_ets()
    rng = np.random.default_rng(N)
    q = rng.uniform(-np.pi, np.pi, (N, 7))
    T, J = ets.fkine_jacob0(q)
    nt.assert_allclose(T.reshape(-1, 4, 4), oracle.fkine(ch, q), atol=TOL)
    nt.assert_allclose(J.reshape(-1, 6, 7), oracle.jacob0(ch, q), atol=TOL)


@pytest.mark.parametrize("coalesced", [1, 0])
def test_store_path_variants_agree(coalesced):
    """run-time-n tile kernel (forced with reg=0), both store paths."""
    rtbhip.tune("coalesced", coalesced)
    rtbhip.tune("reg", 0)
    try:
        ets = rtbhip.models.Panda().ets()
        ch = chains.panda_ets()
        rng = np.random.default_rng(3)
        q = rng.uniform(-np.pi, np.pi, (777, 7))
        T, J = ets.fkine_jacob0(q)
        nt.assert_allclose(T, oracle.fkine(ch, q), atol=TOL)
        nt.assert_allclose(J, oracle.jacob0(ch, q), atol=TOL)
    finally:
        rtbhip.tune("coalesced", 1)
        rtbhip.tune("reg", 1)


def test_huge_and_nonfinite_joint_values():
    """|q| >= 2^20 takes the library sincos path; NaN stays NaN (no hang, no garbage elsewhere)."""
    ets = rtbhip.models.Panda().ets()
    ch = chains.panda_ets()
    rng = np.random.default_rng(8)
    q = rng.uniform(-np.pi, np.pi, (200, 7))
    q[5, 2] = 1.5e7
    q[77, 0] = -3.0e9
    T, J = ets.fkine_jacob0(q)
    nt.assert_allclose(T, oracle.fkine(ch, q), atol=1e-9)
    nt.assert_allclose(J, oracle.jacob0(ch, q), atol=1e-9)
    q[9, 4] = np.nan
    T, J = ets.fkine_jacob0(q)
    assert np.isnan(T[9]).any() and not np.isnan(np.delete(T, 9, axis=0)).any()


@pytest.mark.parametrize("n", [1, 2, 5, 8, 9, 10, 11, 16, 23])
def test_joint_counts(n):
    rng = np.random.default_rng(n)
    axes = ["Rx", "Ry", "Rz", "tx", "ty", "tz"]
    spec = []
    for j in range(n):
        spec.append((axes[rng.integers(6)], float(rng.normal())))
        spec.append((axes[rng.integers(6)], None, bool(rng.integers(2))))
    ets = product_ets(spec)
    ch = chains.Chain(spec)
    q = rng.normal(size=(200, n))
    for frame, fn in ((0, ets.jacob0), (1, ets.jacobe)):
        nt.assert_allclose(fn(q), oracle.jacob(ch, q, frame=frame), atol=1e-9)
    nt.assert_allclose(ets.eval(q), oracle.fkine(ch, q), atol=1e-9)
    if n <= 11:
        nt.assert_allclose(ets.hessian0(q[:20]), oracle.hessian0(ch, q[:20]), atol=1e-9)


def test_config1_puma_dh_fkine_1e3():
    """BASELINE configs[0]: Puma560 6-DOF DH, fkine over 1e3 random q."""
    puma = rtbhip.models.DH.Puma560()
    tab = chains.puma560()
    rng = np.random.default_rng(0)
    q = rng.uniform(tab.qlim[:, 0], tab.qlim[:, 1], (1000, 6))
    T = puma.fkine(q)
    nt.assert_allclose(T, oracle.dh_fkine(tab.dh, 0, q), atol=TOL)       # closed-form DHLink.A product
    nt.assert_allclose(T, oracle.fkine(tab.ets(), q), atol=TOL)          # ETS lowering through fknm's algorithm
    nt.assert_allclose(puma.fkine(REF["puma_q"]), REF["puma_fkine"], atol=TOL)
    nt.assert_allclose(puma.jacob0(REF["puma_q"]), REF["puma_jacob0"], atol=TOL)


def test_G11_dh_literals():
    r0 = rtbhip.DHRobot([rtbhip.PrismaticDH(), rtbhip.RevoluteDH(), rtbhip.PrismaticDH(theta=2.0), rtbhip.RevoluteDH()])
    q = np.array([1, 2, 3, 4.0])
    nt.assert_array_almost_equal(r0.fkine(q), LIT["G11_dh_rprp_fkine"], decimal=6)        # test_DHRobot.py:170-189
    TT = r0.fkine(np.tile(q, (4, 1)))                                                      # :191-208
    for i in range(4):
        nt.assert_array_almost_equal(TT[i], LIT["G11_dh_rprp_fkine"], decimal=6)
    panda = rtbhip.models.DH.Panda()
    nt.assert_array_almost_equal(panda.fkine(np.arange(1, 8.0)), LIT["G11_dh_panda_fkine"], decimal=4)   # :438-451
    r1 = rtbhip.DHRobot([rtbhip.PrismaticDH(theta=4), rtbhip.RevoluteDH(a=2), rtbhip.PrismaticDH(theta=2), rtbhip.RevoluteDH()])
    nt.assert_array_almost_equal(r1.jacobe(q), LIT["G11_dh_rprp_jacobe"], decimal=4)       # :453-479


def test_G9_puma_rne_literals_and_traj_and_reinit():
    puma = rtbhip.models.DH.Puma560()
    z, o = np.zeros(6), np.ones(6)
    qn = puma.qn
    nt.assert_array_almost_equal(puma.rne(qn, z, z), LIT["G9_puma_rne_tr0"], decimal=4)    # test_DHRobot.py:1036-1062
    nt.assert_array_almost_equal(puma.rne(qn, z, o), LIT["G9_puma_rne_tr1"], decimal=4)
    nt.assert_array_almost_equal(puma.rne(qn, o, o), LIT["G9_puma_rne_tr2"], decimal=4)
    nt.assert_array_almost_equal(puma.rne(qn, o, z), LIT["G9_puma_rne_tr3"], decimal=4)
    nt.assert_array_almost_equal(puma.rne(qn, o, o, gravity=[0, 0, 0]), LIT["G9_puma_rne_tr4"], decimal=4)
    nt.assert_array_almost_equal(puma.rne(qn, z, z, fext=LIT["G9_fext"]), LIT["G9_puma_rne_tr5"], decimal=4)
    t = puma.rne(np.c_[qn, qn].T, np.c_[z, o].T, np.c_[z, o].T)                             # :1064-1076
    nt.assert_array_almost_equal(t[0], LIT["G9_puma_rne_tr0"], decimal=4)
    nt.assert_array_almost_equal(t[1], LIT["G9_puma_rne_tr2"], decimal=4)
    puma.delete_rne()                                                                       # :1078-1090
    nt.assert_array_almost_equal(puma.rne(qn, z, z), LIT["G9_puma_rne_tr0"], decimal=4)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def test_rne_reference_run_fixtures():
    puma, pd = rtbhip.models.DH.Puma560(), rtbhip.models.DH.Panda()
    a = (REF["puma_q"], REF["puma_qd"], REF["puma_qdd"])
    assert _rel(puma.rne(*a), REF["puma_rne"]) <= 1e-9
    assert _rel(puma.rne(*a, fext=[1, 2, 3, 1, 2, 3]), REF["puma_rne_fext"]) <= 1e-9
    assert _rel(puma.rne(*a, gravity=[0, 0, 0]), REF["puma_rne_g0"]) <= 1e-9
    assert _rel(puma.rne(*a, gravity=[1.5, -2.0, -9.0]), REF["puma_rne_gx"]) <= 1e-9
    b = (REF["pandadh_q"], REF["pandadh_qd"], REF["pandadh_qdd"])
    assert _rel(pd.rne(*b), REF["pandadh_rne"]) <= 1e-9
    assert _rel(pd.rne(*b, fext=[-1, 0.5, 2, 0.3, -0.2, 0.1]), REF["pandadh_rne_fext"]) <= 1e-9
    # prismatic branches incl. prismatic-first chains, DH and MDH, through the raw C ABI objects
    import ctypes as C
    from rtbhip import _lib
    for name in ("rprp0", "rprp1", "prp0", "prp1"):
        L = np.ascontiguousarray(REF[name + "_L24"])
        n = L.shape[0]
        h = C.c_uint64(0)
        _lib.check(_lib.lib().rtbhip_dyn_create(_lib.host_ptr(L), n, int(name[-1]), C.byref(h)))
        q, qd, qdd = (np.ascontiguousarray(REF[name + s]) for s in ("_q", "_qd", "_qdd"))
        tau = np.empty_like(q)
        g = np.ascontiguousarray(-np.array([0.5, -1.0, -9.81]))
        f = np.array([1, 2, 3, 4, 5, 6.0]) if name.startswith("rprp") else None
        _lib.check(_lib.lib().rtbhip_rne(h.value, _lib.host_ptr(q), _lib.host_ptr(qd), _lib.host_ptr(qdd), q.shape[0],
                                         _lib.host_ptr(g), _lib.host_ptr(f), _lib.host_ptr(tau), 0, None))
        assert _rel(tau, REF[name + "_rne"]) <= 1e-9, name
        _lib.check(_lib.lib().rtbhip_dyn_destroy(h.value))


@pytest.mark.parametrize("n", [1, 3, 8, 9, 14])
def test_rne_link_counts_vs_oracle(n):
    """n <= 8 takes the register-resident template, n > 8 the run-time-n fallback."""
    rng = np.random.default_rng(100 + n)
    for mdh in (0, 1):
        links = []
        for j in range(n):
            kw = dict(a=rng.normal() * 0.2, alpha=rng.normal(), m=abs(rng.normal()) + 0.1, r=rng.normal(size=3) * 0.1,
                      I=np.abs(rng.normal(size=3)) * 0.1, Jm=1e-4, G=5.0 + j, B=1e-3, Tc=[0.1, -0.1])
            pris = rng.integers(3) == 0
            cls = {(0, False): rtbhip.RevoluteDH, (0, True): rtbhip.PrismaticDH,
                   (1, False): rtbhip.RevoluteMDH, (1, True): rtbhip.PrismaticMDH}[(mdh, bool(pris))]
            links.append(cls(theta=rng.normal(), **kw) if pris else cls(d=rng.normal() * 0.2, **kw))
        rob = rtbhip.DHRobot(links, gravity=[0.3, -0.2, -9.81])
        q, qd, qdd = rng.normal(size=(130, n)), rng.normal(size=(130, n)), rng.normal(size=(130, n))
        tau = rob.rne(q, qd, qdd, fext=[1, -2, 3, 0.1, 0.2, -0.3])
        ref = oracle.rne_dh(rob.L24(), mdh, q, qd, qdd, -rob.gravity, [1, -2, 3, 0.1, 0.2, -0.3])
        assert _rel(tau, ref) <= 1e-9


def test_device_tensor_path_matches_host_path():
    ets = rtbhip.models.Panda().ets()
    rng = np.random.default_rng(11)
    q = rng.uniform(-np.pi, np.pi, (5000, 7))
    Th, Jh = ets.fkine_jacob0(q)
    qd = torch.from_numpy(q).cuda()
    Td, Jd = ets.fkine_jacob0(qd)
    assert Td.is_cuda and Td.shape == (5000, 4, 4) and Jd.shape == (5000, 6, 7)
    torch.cuda.synchronize()
    nt.assert_array_equal(Td.cpu().numpy(), Th)
    nt.assert_array_equal(Jd.cpu().numpy(), Jh)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                      # launches follow torch's current stream
        T2 = ets.eval(qd)
    s.synchronize()
    nt.assert_array_equal(T2.cpu().numpy(), Th)


def test_fleet_one_launch_many_chains():
    rng = np.random.default_rng(21)
    panda = rtbhip.models.Panda().ets()
    puma = rtbhip.models.DH.Puma560().ets()
    mixed = product_ets(mixed_spec())
    chs = [panda, puma, mixed, panda]
    och = [chains.panda_ets(), chains.puma560().ets(), chains.Chain(mixed_spec()), chains.panda_ets()]
    qs = [rng.normal(size=(N, c.n)) for N, c in zip((1000, 65, 1, 129), chs)]
    Ts, Js = rtbhip.fleet_fkine_jacob(chs, qs)
    for T, J, oc, q in zip(Ts, Js, och, qs):
        nt.assert_allclose(T, oracle.fkine(oc, q), atol=TOL)
        nt.assert_allclose(J, oracle.jacob0(oc, q), atol=TOL)


def test_full_size_1e6_properties_and_sampled_parity():
    """BASELINE configs[1] at full size: N = 1e6, device resident.  Size-independent properties +
    oracle parity on a strided sample."""
    N = full_size(1000000)
    ets = rtbhip.models.Panda().ets()
    ch = chains.panda_ets()
    rng = np.random.default_rng(0)
    qh = rng.uniform(-np.pi, np.pi, (N, 7))
    q = torch.from_numpy(qh).cuda()
    T, J0 = ets.fkine_jacob0(q)
    _, Je = ets.fkine_jacob0(q, frame=1)
    R = T[:, :3, :3]
    eye = torch.eye(3, dtype=torch.float64, device=DEV())
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-13                      # rotations stay orthonormal
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-13
    assert bool((T[:, 3, :] == torch.tensor([0, 0, 0, 1.0], dtype=torch.float64, device=DEV())).all())
    # jacobe == blkdiag(R^T, R^T) jacob0 for every configuration
    Jv = R.transpose(1, 2) @ J0[:, :3, :]
    Jw = R.transpose(1, 2) @ J0[:, 3:, :]
    assert float((torch.cat([Jv, Jw], dim=1) - Je).abs().max()) < 1e-12
    # Jacobian is the derivative of fkine: central difference on joint 3 for a strided sample
    idx = torch.arange(0, N, 9973, device=DEV())
    h = 1e-6
    dq = torch.zeros(7, dtype=torch.float64, device=DEV()); dq[3] = h
    Tp, Tm = ets.eval(q[idx] + dq), ets.eval(q[idx] - dq)
    assert float(((Tp[:, :3, 3] - Tm[:, :3, 3]) / (2 * h) - J0[idx, :3, 3]).abs().max()) < 1e-7
    # idempotence: a second launch writes bit-identical output
    T2, J2 = ets.fkine_jacob0(q)
    assert bool((T2 == T).all()) and bool((J2 == J0).all())
    sel = np.arange(0, N, 997)
    nt.assert_allclose(T[sel].cpu().numpy(), oracle.fkine(ch, qh[sel]), atol=TOL)
    nt.assert_allclose(J0[sel].cpu().numpy(), oracle.jacob0(ch, qh[sel]), atol=TOL)


def test_full_size_rne_1e6_sampled_parity_and_linearity():
    """BASELINE configs[3] per-GPU share (1.25e6 of the 1e7 triples), DH Panda."""
    N = full_size(1250000)
    pd = rtbhip.models.DH.Panda()
    tab = chains.panda_dh()
    rng = np.random.default_rng(3)
    qh = rng.uniform(tab.qlim[:, 0], tab.qlim[:, 1], (N, 7))
    qdh, qddh = rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
    q, qd, qdd = (torch.from_numpy(x).cuda() for x in (qh, qdh, qddh))
    tau = pd.rne(q, qd, qdd)
    sel = np.arange(0, N, 1249)
    ref = oracle.rne_dh(tab.L24(), 1, qh[sel], qdh[sel], qddh[sel], -tab.gravity)
    assert _rel(tau[sel].cpu().numpy(), ref) <= 1e-9
    # inverse dynamics is affine in qdd at fixed (q, qd): tau(qdd1+qdd2) - tau(qdd1) == tau(qdd2) - tau(0)
    z = torch.zeros_like(qdd)
    lhs = pd.rne(q, qd, qdd + 1.0) - tau
    rhs = pd.rne(q, qd, z + 1.0) - pd.rne(q, qd, z)
    assert float((lhs - rhs).abs().max()) < 1e-9


# ---------------------------------------------------------------- inverse kinematics on the device
def _panda_limited():
    ets = rtbhip.models.Panda().ets()
    ets.qlim = chains.PANDA_QLIM
    return ets, chains.panda_ets(with_limits=True)


def test_ik_G10_single_target_tuple():
    """test_IK.py:632-708: chan / wampler k=0.01 / sugihara k=0.01; success and E(FK(q), Tep) < 1e-5."""
    ets, ch = _panda_limited()
    Tep = oracle.fkine(ch, np.array([0, -0.3, 0, -2.2, 0, 2, np.pi / 4]))[0]
    for method, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):
        q, ok, it, se, E = ets.ik_LM(Tep, k=k, method=method)
        assert q.shape == (7,) and ok == 1 and isinstance(it, int) and E < 1e-6
        e = oracle.angle_axis(oracle.fkine(ch, q)[0], Tep)
        assert 0.5 * e @ e < 1e-5
        sol = ets.ikine_LM(Tep, k=k, method=method)                              # test_IK.py:451-492
        assert sol.success and sol.q.shape == (7,) and sol.residual < 1e-6
        e = oracle.angle_axis(oracle.fkine(ch, sol.q)[0], Tep)
        assert 0.5 * e @ e < 1e-5


@pytest.mark.parametrize("flavour", [0, 1])
def test_ik_equals_oracle_given_same_restarts(flavour):
    ets, ch = _panda_limited()
    rng = np.random.default_rng(23)
    N = 96
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    for method, k in (("chan", 1.0), ("sugihara", 0.01)):
        single, q, ok, it, se, E = ets._ik(Tep, None, 30, 100, 1e-6, None, True, k, method, flavour, 1234)
        for i in range(N):
            rs = np.array([ets.ik_restart(1234, i, d) for d in range(101)])
            o = oracle.ik_lm(ch, Tep[i], k=k, method=method, restarts=rs) if flavour == 0 else \
                oracle.ikine_lm(ch, Tep[i], rs[:100], k=k, method=method)
            assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i]), (i, o[1:4], ok[i], it[i], se[i])
            nt.assert_allclose(q[i], o[0], atol=1e-6)


def test_ik_reference_run_fixtures_first_search():
    ets, _ = _panda_limited()
    for method, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):
        q, ok, it, se, E = ets.ik_LM(REF["ik_Tep"], q0=REF["ik_q0"], k=k, method=method)
        meta = REF["ik_%s_meta" % method]
        first = (meta[:, 2] == 1) & (meta[:, 0] == 1)
        nt.assert_array_equal(np.c_[ok, it, se][first], meta[first])
        nt.assert_allclose(q[first], REF["ik_%s_q" % method][first], atol=1e-6)


def test_ik_gn_nr_reference_fixtures_and_statistics():
    """IK_GN_c / IK_NR_c through the same device scheduler (method 3 / 4): first-search cases equal the
    reference's own outputs; over random reachable targets every reported success satisfies E < tol and
    the limits, and the success rate is not below the reference's own on its fixture sample."""
    ets, ch = _panda_limited()
    puma = rtbhip.models.DH.Puma560().ets()
    puma.qlim = chains.puma560().qlim.T
    for key, e, Tep, q0, fn, kw in (("ik_gn", ets, REF["ik_Tep"], REF["ik_q0"], "ik_GN", {}),
                                    ("ik_nr", ets, REF["ik_Tep"], REF["ik_q0"], "ik_NR", {}),
                                    ("ik_nrd", ets, REF["ik_Tep"], REF["ik_q0"], "ik_NR", dict(pinv_damping=0.05)),
                                    ("ik6_gn", puma, REF["ik6_Tep"], REF["ik6_q0"], "ik_GN", dict(pinv=False)),
                                    ("ik6_nr", puma, REF["ik6_Tep"], REF["ik6_q0"], "ik_NR", dict(pinv=False))):
        q, ok, it, se, E = getattr(e, fn)(Tep, q0=q0, **kw)
        meta = REF[key + "_meta"]
        first = (meta[:, 2] == 1) & (meta[:, 0] == 1)
        nt.assert_array_equal(np.c_[ok, it, se][first], meta[first])
        nt.assert_allclose(q[first], REF[key + "_q"][first], atol=1e-6)
    rng = np.random.default_rng(12)
    N = 5000
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    for fn, ref_rate in (("ik_GN", REF["ik_gn_meta"][:, 0].mean()), ("ik_NR", REF["ik_nr_meta"][:, 0].mean())):
        q, ok, it, se, E = getattr(ets, fn)(Tep, seed=5)
        good = ok == 1
        assert good.mean() >= min(0.9, ref_rate) - 0.03
        assert np.all(E[good] < 1e-6)
        assert np.all(q[good] >= ch.qlim[0] - 1e-12) and np.all(q[good] <= ch.qlim[1] + 1e-12)
        err = np.abs(oracle.fkine(ch, q[good]) - Tep[good]).reshape(good.sum(), -1).max(axis=1)
        assert err.max() < 2e-3                       # E = e.e/2 < 1e-6  =>  |e| < 1.5e-3


def test_ikine_nr_gn_python_flavour():
    """ETS.ikine_NR / ikine_GN: IKSolution results, success on the reference's G10 target, pinv rule for redundant arms."""
    ets, ch = _panda_limited()
    Tep = oracle.fkine(ch, np.array([0, -0.3, 0, -2.2, 0, 2.0, np.pi / 4]))[0]
    for fn in (ets.ikine_NR, ets.ikine_GN):
        sol = fn(Tep, pinv=True, seed=0)
        assert sol.success and sol.residual < 1e-6                      # reference tests/test_IK.py:19-37, 253-273
        e = oracle.angle_axis(oracle.fkine(ch, sol.q)[0], Tep)
        assert 0.5 * e @ e < 1e-5
        bad = fn(Tep, slimit=5)                                         # pinv=False on a 7-joint arm: numpy.linalg.inv raises in every
        assert not bad.success and bad.iterations == 5 and bad.searches == 5 and "LinAlgError" in bad.reason   # search (robot/IK.py:320-323)
    sol = ets.ikine_NR(np.stack([Tep, Tep]), pinv=True, seed=1)
    assert sol.q.shape == (2, 7) and sol.each["success"].all()


def test_ikine_nullspace_terms():
    """kq / km / ps / pi of the Python solvers (robot/IK.py:507-576).  The reference's own cases: tests/test_IK.py:166-183
    (IK_NR pinv kq=0.01 km=1), :194-205 (IK_LM chan kq=km=0.1), :261-272 (IK_GN pinv kq=km=1) -- success and E < 1e-5 on
    the Panda target; then first-search agreement with the NumPy restatement on a batch with active limit avoidance."""
    ets, ch = _panda_limited()
    plain = rtbhip.models.Panda().ets()                    # the reference's tests use the model without Franka limits
    Tep = oracle.fkine(ch, np.array([0, -0.3, 0, -2.2, 0, 2.0, np.pi / 4]))[0]
    for e in (plain, ets):
        for sol in (e.ikine_NR(Tep, pinv=True, seed=0, kq=0.01, km=1.0), e.ikine_LM(Tep, method="chan", seed=0, kq=0.1, km=0.1),
                    e.ikine_GN(Tep, pinv=True, seed=0, kq=1.0, km=1.0)):
            assert sol.success and sol.residual < 1e-6
            err = oracle.angle_axis(oracle.fkine(ch, sol.q)[0], Tep)
            assert 0.5 * err @ err < 1e-5
    rng = np.random.default_rng(41)
    N = 200
    qs = rng.uniform(ch.qlim[0] + 0.25, ch.qlim[1] - 0.25, (N, 7))
    qs[::2, 3] = ch.qlim[1, 3] - 0.12
    qs[1::4, 1] = ch.qlim[0, 1] + 0.1
    T = oracle.fkine(ch, qs)
    q0 = np.clip(qs + 0.03 * rng.normal(size=qs.shape), ch.qlim[0] + 0.02, ch.qlim[1] - 0.02)
    for step, kw, ns in (("lm", dict(method="chan", k=1.0), (0.1, 0.1, 0.0, 0.3)), ("nr", dict(pinv=True), (0.01, 1.0, 0.0, 0.3)),
                         ("lm", dict(method="wampler", k=0.01), (0.5, 0.0, 0.05, 0.4))):
        kq, km, ps, pi = ns
        fn = ets.ikine_LM if step == "lm" else ets.ikine_NR
        sol = fn(T, q0=q0, seed=5, slimit=3, kq=kq, km=km, ps=ps, pi=pi, **kw)
        base = fn(T, q0=q0, seed=5, slimit=3, **kw)
        assert np.nanmax(np.abs(sol.q - base.q)) > 1e-6                 # the term acts
        checked = 0
        for i in range(0, N, 4):
            rs = np.array([q0[i]] + [ets.ik_restart(5, i, d) for d in range(1, 3)])
            o = oracle.ikine_py(ch, T[i], rs, step=step, slimit=3, k=kw.get("k", 0.0), kq=kq, km=km, ps=ps, pi=pi, method=kw.get("method", "chan"))
            if o[1] and o[3] == 1:
                checked += 1
                assert (o[1], o[2], o[3]) == (sol.each["success"][i], sol.each["iterations"][i], sol.each["searches"][i])
                nt.assert_allclose(sol.q[i], o[0], atol=1e-6)
        assert checked >= 25
    # a device batch gives what the host path gives; chains outside 6..8 joints are refused loudly (nothing dropped silently)
    import torch
    st = ets.ikine_LM(torch.from_numpy(T).cuda(), q0=torch.from_numpy(q0).cuda(), seed=5, slimit=3, kq=0.1, km=0.1)
    sh = ets.ikine_LM(T, q0=q0, seed=5, slimit=3, kq=0.1, km=0.1)
    nt.assert_array_equal(st.q, sh.q)
    long13 = rtbhip.DHRobot([rtbhip.RevoluteDH(a=0.1, d=0.05, alpha=[0.0, 1.5][k % 2]) for k in range(13)]).ets()
    # null-space step variants are built in for 6..12 joints; beyond, the same template is instantiated at run time (csrc/jit.cpp) -- the
    # reference's loop takes any n (robot/IK.py:542-576).  On the device: served, and the solution reaches its target.
    from helpers import replaying
    if not replaying():
        from rtbhip import jit
        q13 = np.full(13, 0.2)
        if jit.stats()["available"]:
            s13 = long13.ikine_LM(long13.eval(q13), q0=q13 + 0.02, kq=0.1, km=0.1, seed=1)
            assert s13.success and np.abs(np.asarray(long13.eval(s13.q)) - np.asarray(long13.eval(q13))).max() < 1e-4
        else:
            with pytest.raises(rtbhip.RtbHipError):
                long13.ikine_LM(long13.eval(q13), kq=0.1)
    five = urdf.load("px100").ets()
    assert five.n < 6
    with pytest.raises(rtbhip.RtbHipError):
        five.ikine_LM(five.eval(np.zeros(five.n)), kq=0.1)
    assert five.ikine_LM(five.eval(np.full(five.n, 0.1)), q0=np.full(five.n, 0.12), km=0.1).success     # km alone: the reference's guard drops it
    with pytest.raises(rtbhip.RtbHipError):
        lib_ik_flavour0_nullspace(ets, Tep)


def lib_ik_flavour0_nullspace(ets, Tep):
    """kq with the C-solver flavour is an argument error at the ABI."""
    from rtbhip._lib import lib, check, host_ptr, MEM_HOST
    q = np.empty((1, 7)); ok = np.empty(1, np.int32); it = np.empty(1, np.int32); se = np.empty(1, np.int32); E = np.empty(1)
    T = np.ascontiguousarray(Tep.reshape(1, 4, 4))
    check(lib().rtbhip_ik_lm_nullspace(ets._handle(), host_ptr(T), 1, None, 30, 100, 1e-6, 1, None, 1.0, 0, 0, 0, 0.1, 0.0, 0.0, None,
                                       host_ptr(q), host_ptr(ok), host_ptr(it), host_ptr(se), host_ptr(E), MEM_HOST, None))


def test_ik_config3_1e5_targets_statistics():
    """BASELINE configs[2]: 1e5 random reachable targets, Franka limits, defaults
    (ilimit 30, slimit 100, tol 1e-6, chan, k=1).  Every reported success must satisfy E < tol,
    reproduce the target pose to 1e-5 and respect the joint limits; the success rate must not be
    below the CPU oracle's (same algorithm, same generator) minus 0.1 %."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(1)
    N = full_size(100000, 10)
    qs = rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7))
    Tep = torch.from_numpy(oracle.fkine(ch, qs)).cuda()
    q, ok, it, se, E = ets.ik_LM(Tep, seed=2)
    torch.cuda.synchronize()
    qh, okh, Eh = q.cpu().numpy(), ok.cpu().numpy().astype(bool), E.cpu().numpy()
    assert okh.mean() > 0.99
    assert (Eh[okh] < 1e-6).all()
    assert (qh[okh] >= ch.qlim[0]).all() and (qh[okh] <= ch.qlim[1]).all()
    Tsol = ets.eval(q).cpu().numpy()
    Th = Tep.cpu().numpy()
    # the reference's own criterion (tests/test_IK.py:15): the error of the REACHED pose, recomputed from FK(q), below 1e-5 --
    # here at the solver's own level: E(FK(q), Tep) within rounding of the reported residual, hence < 1e-6
    e = rtbhip.angle_axis(torch.from_numpy(Tsol).cuda(), Tep).cpu().numpy()
    Ecalc = 0.5 * (e * e).sum(axis=1)
    assert (Ecalc[okh] < 1e-6).all()
    assert np.abs(Ecalc[okh] - Eh[okh]).max() < 1e-12
    assert np.abs(Tsol[okh] - Th[okh]).max() < 2.0 * np.sqrt(2e-6)      # |dT| <= |e| (1 + reach) with |e| < sqrt(2 tol)
    sub = np.arange(0, N, 250)
    o_ok = []
    for i in sub:
        rs = np.array([ets.ik_restart(2, int(i), d) for d in range(101)])
        o = oracle.ik_lm(ch, Th[i], restarts=rs)
        o_ok.append(o[1])
        assert (o[1], o[2], o[3]) == (ok[i].item(), it[i].item(), se[i].item())
    assert okh[sub].mean() >= np.mean(o_ok) - 1e-3
    assert 0.985 < okh.mean() < 0.997                                   # 0.9 % of this batch exhausts all 100 searches (in the oracle too)
    # failures (if any) carry the reference's bookkeeping: searches == slimit + 1
    if (~okh).any():
        assert (se.cpu().numpy()[~okh] == 101).all()


def test_ik_ten_joint_chain_equals_oracle():
    """URDF Fetch (10 joints on the path): IK at one wave per SIMD; same restart vectors -> the oracle's results."""
    from rtbhip import urdf
    from helpers import chain_from_ets
    ets = urdf.load("Fetch").ets()
    ets.qlim = np.clip(ets.qlim, -np.pi, np.pi)
    ch = chain_from_ets(ets)
    rng = np.random.default_rng(10)
    N = 300
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 10)))
    q, ok, it, se, E = ets.ik_LM(Tep, seed=5, slimit=30)
    for i in range(0, N, 15):
        rs = np.array([ets.ik_restart(5, i, d) for d in range(31)])
        o = oracle.ik_lm(ch, Tep[i], restarts=rs, slimit=30)
        assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
        nt.assert_allclose(q[i], o[0], atol=1e-6)
    good = ok == 1
    assert good.mean() > 0.8 and np.all(E[good] < 1e-6)
    seventeen = rtbhip.DHRobot([rtbhip.RevoluteDH(a=0.1) for _ in range(17)]).ets()
    from helpers import large_sizes_served
    if large_sizes_served():                   # 17 joints: no built-in k_ik -- instantiated at run time (tests/test_large_chains_gpu.py checks it against the oracle)
        q17 = np.linspace(0.1, 0.5, 17)
        s17 = seventeen.ik_LM(seventeen.eval(q17), q0=q17 + 0.01)
        assert s17[1] == 1
    else:
        with pytest.raises(rtbhip.RtbHipError):
            seventeen.ik_LM(np.eye(4))


def test_ik_fourteen_joint_chain_equals_oracle():
    """Chains of 13..16 joints (the normal equations spill to scratch): a 14-joint arm -- two stacked 7-joint Panda-like
    halves -- must walk exactly the oracle's sequential loops, C and Python flavour; null-space terms on 9 and 10 joints
    (Gen3 + finger, Fetch) run and converge."""
    from helpers import chain_from_ets
    ET = rtbhip.ET
    half = lambda: (ET.tz(0.2) * ET.Rz() * ET.Rx(-np.pi / 2) * ET.Rz() * ET.Rx(np.pi / 2) * ET.tz(0.2) * ET.Rz() * ET.tx(0.05)
                    * ET.Rx(np.pi / 2) * ET.Rz() * ET.tx(-0.05) * ET.Rx(-np.pi / 2) * ET.tz(0.2) * ET.Rz() * ET.Rx(np.pi / 2) * ET.Rz()
                    * ET.tx(0.05) * ET.Rx(np.pi / 2) * ET.Rz())
    ets = half() * half()
    assert ets.n == 14
    ch = chain_from_ets(ets)
    rng = np.random.default_rng(14)
    N = 40
    qs = rng.uniform(-2.5, 2.5, (N, 14))
    Tep = oracle.fkine(ch, qs)
    for flavour, fn in ((0, ets.ik_LM), (1, None)):
        if flavour == 0:
            q, ok, it, se, E = fn(Tep, seed=5, slimit=12)
        else:
            sol = ets.ikine_LM(Tep, seed=5, slimit=12)
            q, ok, it, se, E = sol.q, sol.each["success"].astype(int), sol.each["iterations"], sol.each["searches"], sol.each["residual"]
        for i in range(0, N, 3):
            rs = np.array([ets.ik_restart(5, i, d) for d in range(13)])
            o = oracle.ik_lm(ch, Tep[i], restarts=rs, slimit=12) if flavour == 0 else oracle.ikine_lm(ch, Tep[i], rs[:12], slimit=12)
            assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
            nt.assert_allclose(q[i], o[0], atol=1e-6)
        assert np.mean(ok) > 0.9
    from rtbhip import urdf as U
    # Null-space terms on 9 / 10 joints.  Gains of 1/kq = 10 push many searches out of the limits and into overflow (the Fetch's
    # prismatic torso with the manipulability term: every search); what the reference's formulas do there is chaotic, so the
    # yardstick is the NumPy restatement target by target, and strict equality is asked of the searches that converge
    # (measured on the MI355X: Gen3 28 of 30 tuples equal, 9 first-search successes; Fetch with kq alone 13 of 30, 7).
    for name, kw, min_agree in (("KinovaGen3", {"kq": 0.1, "km": 0.1}, 25), ("Fetch", {"kq": 0.1}, 0)):
        e = U.load(name).ets()
        assert e.n in (9, 10)
        e.qlim = np.clip(e.qlim, -np.pi, np.pi)
        c2 = chain_from_ets(e)
        span = c2.qlim[1] - c2.qlim[0]
        q2 = rng.uniform(c2.qlim[0] + 0.1 * span, c2.qlim[1] - 0.1 * span, (30, e.n))
        T2 = oracle.fkine(c2, q2)
        q0 = np.clip(q2 + 0.02 * span * rng.normal(size=q2.shape), c2.qlim[0], c2.qlim[1])
        sol = e.ikine_LM(T2, q0=q0, seed=1, slimit=5, **kw)
        base = e.ikine_LM(T2, q0=q0, seed=1, slimit=5)
        assert np.nanmax(np.abs(np.nan_to_num(sol.q) - np.nan_to_num(base.q))) > 1e-7      # the terms act
        agree = hits = 0
        for i in range(30):
            o = oracle.ikine_py(c2, T2[i], np.array([q0[i]] + [e.ik_restart(1, i, d) for d in range(1, 5)]), step="lm", slimit=5, **kw)
            same = (o[1], o[2], o[3]) == (sol.each["success"][i], sol.each["iterations"][i], sol.each["searches"][i])
            agree += same
            if o[1] and o[3] == 1:
                hits += 1
                assert same
                nt.assert_allclose(sol.q[i], o[0], atol=1e-6)
        assert agree >= min_agree and hits >= 3


@pytest.mark.parametrize("flavour", [0, 1])
def test_ik_cross_wave_sharing_and_phased_schedule_equal_plain(flavour):
    """The flat schedule ("ik_flat", the default for batches resident at once), cross-wave sharing of search ranges (rtbhip_tune "ik_share") and the phased schedule ("ik_phased") only change WHO runs a
    search: every output must be bit-equal to the plain scheduler's, for reachable and unreachable targets, with and without q0,
    at a size where the grid is under-filled (donations happen) and at one where it is over-filled."""
    import torch
    ets, ch = _panda_limited()
    rng = np.random.default_rng(77 + flavour)
    for N, with_q0, slimit in ((3000, False, 100), (150000, False, 100), (4000, True, 37), (500, False, 9)):
        N = full_size(N, 50)              # (the CPU replay runs ONE schedule whatever the knobs say: there only the plumbing is under test)
        Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
        Tep[::41, :3, 3] += 2.5
        q0 = rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)) if with_q0 else None
        Tt = torch.from_numpy(Tep).cuda()
        q0t = None if q0 is None else torch.from_numpy(q0).cuda()
        run = (lambda: ets.ik_LM(Tt, q0=q0t, seed=3, slimit=slimit)) if flavour == 0 else \
              (lambda: tuple(torch.as_tensor(x) for x in (lambda s: (s.q, s.each["success"], s.each["iterations"], s.each["searches"], s.each["residual"]))(ets.ikine_LM(Tt, q0=q0t, seed=3, slimit=slimit))))
        try:
            rtbhip.tune("ik_share", 0); rtbhip.tune("ik_phased", 0); rtbhip.tune("ik_flat", 0)
            base = [x.cpu().numpy() for x in run()]
            for l0, length in ((4, 8), (8, 8), (1, 3), (7, 40)):      # the flat schedule, four cuts of the search range
                rtbhip.tune("ik_flat", 2); rtbhip.tune("ik_flat_l0", l0); rtbhip.tune("ik_flat_len", length)
                flat = [x.cpu().numpy() for x in run()]
                for a, b in zip(base, flat):
                    nt.assert_array_equal(a, b)
            rtbhip.tune("ik_flat", 0); rtbhip.tune("ik_flat_l0", 0); rtbhip.tune("ik_flat_len", 0)
            rtbhip.tune("ik_share", 2)
            shared = [x.cpu().numpy() for x in run()]
            rtbhip.tune("ik_donate_after", 0)                     # ranges cut as soon as a wave waits, not after three failures
            shared0 = [x.cpu().numpy() for x in run()]
            rtbhip.tune("ik_donate_after", 3)
            rtbhip.tune("ik_share", 0); rtbhip.tune("ik_phased", 2)
            phased = [x.cpu().numpy() for x in run()]
        finally:
            rtbhip.tune("ik_share", 0); rtbhip.tune("ik_phased", 0); rtbhip.tune("ik_donate_after", 3)
            rtbhip.tune("ik_flat", 1); rtbhip.tune("ik_flat_l0", 0); rtbhip.tune("ik_flat_len", 8)          # the defaults (l0 0 = automatic)
        for a, b, b0, c in zip(base, shared, shared0, phased):
            nt.assert_array_equal(a, b)
            nt.assert_array_equal(a, b0)
            nt.assert_array_equal(a, c)
        assert 0 < base[1].sum() < N


def test_ik_kernel_specialisations_agree():
    """The default mask (all ones) and an all-revolute chain without flips run their own kernel instantiations (no products with the weights;
    a straight-line FK walk): same decisions, iterations and searches as the general kernels, q within the solver's own tolerance (a redundant
    arm's solution moves by ~1e-7 along its null space under last-bit differences of the normal equations).  A weighted mask and a chain with a
    prismatic joint must not be affected by the switches at all."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(91)
    N = full_size(20000, 100)             # (under the CPU replay all switches run the one replay: only the plumbing is under test there)
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    outs = {}
    try:
        for u, pl in ((0, 0), (1, 0), (1, 1)):
            rtbhip.tune("ik_unit_we", u); rtbhip.tune("ik_plain", pl)
            outs[(u, pl)] = ets.ik_LM(Tep, seed=4)
            outs[(u, pl, "mask")] = ets.ik_LM(Tep[:min(2000, N)], seed=4, mask=[1, 1, 1, 0.5, 0.5, 0])
    finally:
        rtbhip.tune("ik_unit_we", 1); rtbhip.tune("ik_plain", 1)
    base = outs[(0, 0)]
    for key in ((1, 0), (1, 1)):
        o = outs[key]
        for k in (1, 2, 3):
            nt.assert_array_equal(o[k], base[k])
        ok = base[1] == 1
        # q: the two kernels stop at the same iteration with E equal to ~1e-12, but a redundant arm's iterate may sit ~1e-6 apart along the
        # null space (one target in 20000 does); what they must agree on is the pose they reach
        assert np.abs(o[0][ok] - base[0][ok]).max() < 1e-5 and np.abs(o[4][ok] - base[4][ok]).max() < 1e-9
        worst = np.argsort(np.abs(o[0] - base[0]).max(axis=1) * ok)[-50:]
        assert np.abs(oracle.fkine(ch, o[0][worst]) - oracle.fkine(ch, base[0][worst])).max() < 1e-6
        for a, b in zip(outs[key + ("mask",)], outs[(0, 0, "mask")]):
            nt.assert_array_equal(a, b)                      # a weighted mask never takes the unit-weight kernels
    assert 0.9 < base[1].mean() < 1.0


def test_ik_structure_signature_kernel_returns_the_general_kernel_s_bits():
    """The Panda's chain has a structure signature k_ik is instantiated for (six quarter turns about x, translations on some axes, the flange
    rotation: csrc/ik_kernels.hip kIkSigPandaETS; its constant segments are then multiplied in the form of their class, csrc/kin_device.h).
    Every structured product is the general product with its exact zeros dropped, operation for operation: q, E, success, iterations and
    searches must be BIT-IDENTICAL to the general kernel's (rtbhip_tune "ik_sig" = 0), in both solver flavours and with a weighted mask (which
    never takes the specialised kernel).  A chain with another signature is not affected by the switch."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(92)
    N = full_size(20000, 100)
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    other = rtbhip.ETS(list(rtbhip.models.Panda().ets())[:-1] + [rtbhip.ET.Rx(0.2)])          # the flange replaced: another signature
    other.qlim = ets.qlim
    outs = {}
    try:
        for sig in (1, 0):
            rtbhip.tune("ik_sig", sig)
            outs[sig] = [ets.ik_LM(Tep, seed=5), ets.ikine_LM(Tep[:min(N, 3000)], seed=5), ets.ik_LM(Tep[:min(N, 2000)], seed=5, mask=[1, 1, 1, 0.5, 0.5, 0]),
                         other.ik_LM(Tep[:min(N, 2000)], seed=5)]
    finally:
        rtbhip.tune("ik_sig", 1)
    for a, b in zip(outs[1], outs[0]):
        a = a if isinstance(a, tuple) else (a.q, a.success, a.iterations, a.searches, a.residual)
        b = b if isinstance(b, tuple) else (b.q, b.success, b.iterations, b.searches, b.residual)
        for x, y in zip(a, b):
            nt.assert_array_equal(np.asarray(x), np.asarray(y))
    assert 0.9 < np.asarray(outs[1][0][1]).mean() <= 1.0          # (under the CPU replay the sample is 200 targets: all may succeed)
    # the other robots with an instantiation: the Panda read from its URDF, a UR5 (six joints)
    for robot, end in (("Panda", None), ("UR5", "tool0")):
        e = urdf.load(robot).ets(end=end)
        lim = np.clip(e.qlim, -2.8, 2.8)
        T = np.asarray(e.eval(np.random.default_rng(93).uniform(lim[0], lim[1], (min(N, 4000), e.n))))
        res = {}
        try:
            for sig in (1, 0):
                rtbhip.tune("ik_sig", sig)
                res[sig] = e.ik_LM(T, seed=6)
        finally:
            rtbhip.tune("ik_sig", 1)
        # every structured segment form is the general product's operation sequence with its exact zeros and ones rewritten (kin_device.h: dotk,
        # explicit fused multiply-adds under fp contract(off)), and everything downstream of it in the iteration -- the joint rotations, the Jacobian's
        # cross products, the pose error, the normal equations, the LDL^T solve -- is written out the same way (kin_device.h: mix_pp; what the compiler
        # makes of `a c + b s` depends on where a and b came from, scripts/contraction_probe.hip): the SAME BITS by construction -- the Panda's and
        # (round 5: 6e-9 apart) the UR's too
        for x, y in zip(res[1], res[0]):
            nt.assert_array_equal(np.asarray(x), np.asarray(y))
        assert np.asarray(res[1][1]).mean() > 0.8


@pytest.mark.parametrize("shape", ["six revolute", "seven with a flipped joint", "six with a prismatic joint", "eight revolute", "five revolute"])
def test_ik_kernel_variants_on_other_chain_shapes_equal_the_oracle(shape):
    """Which k_ik instantiation serves a call depends on the chain (all-revolute without flips: the straight-line walk; anything else: the
    general one) and on the mask (all ones: no products with the weights).  Random chains of each kind, with the default mask and with a
    weighted one, from a supplied start (no generator involved): (success, iterations, searches) equal the C restatement of the reference's
    loop wherever it converges in its first search, q within 1e-6."""
    rng = np.random.default_rng({"six revolute": 1, "seven with a flipped joint": 2, "six with a prismatic joint": 3, "eight revolute": 4, "five revolute": 5}[shape])
    n = {"six revolute": 6, "seven with a flipped joint": 7, "six with a prismatic joint": 6, "eight revolute": 8, "five revolute": 5}[shape]
    axes = ["Rz", "Ry", "Rx"]
    spec = []
    for j in range(n):
        ax = axes[int(rng.integers(0, 3))]
        flip = shape == "seven with a flipped joint" and j == 3
        if shape == "six with a prismatic joint" and j == 2:
            ax = "tz"
        spec.append((ax, None, flip))
        spec.append((["tx", "ty", "tz"][int(rng.integers(0, 3))], float(rng.uniform(0.1, 0.4))))
        if j % 2 == 0:
            spec.append((["Rx", "Ry"][int(rng.integers(0, 2))], float(rng.uniform(-1.5, 1.5))))
    lo = np.array([-0.3 if s[0].startswith("t") else -2.6 for s in spec if s[1] is None])
    qlim = np.array([lo, -lo * np.where(lo > -1, 2.0, 1.0)])
    ets = product_ets(spec, qlim=qlim)
    ch = chains.Chain(spec, qlim=qlim)
    N = 300
    qs = rng.uniform(qlim[0] * 0.9, qlim[1] * 0.9, (N, n))
    Tep = oracle.fkine(ch, qs)
    q0 = np.clip(qs + 0.2 * rng.normal(size=qs.shape), qlim[0], qlim[1])
    masks = [None, [1, 1, 1, 0.5, 0.5, 0.25]] if n >= 6 else [[1, 1, 1, 0, 0, 1], [1, 0.5, 1, 0, 0, 1]]
    for mask in masks:
        q, ok, it, se, E = ets.ik_LM(Tep, q0=q0, mask=mask, slimit=3, seed=7)
        checked = 0
        for i in range(N):
            o = oracle.ik_lm(ch, Tep[i], q0=q0[i], restarts=np.zeros((2, n)), slimit=1, we=None if mask is None else np.array(mask, dtype=float))
            if o[1] and o[3] == 1:
                checked += 1
                assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i]), (shape, mask, i)
                nt.assert_allclose(q[i], o[0], atol=1e-6)
        assert checked >= N // 3, (shape, mask, checked)


def test_ik_qp_error_paths_and_batch_vs_oracle():
    """ETS.ikine_QP (IK_QP, robot/IK.py:1222-1520): the manipulability term / the joint-limit rows on a 5-joint arm and ps == pi
    are refused loudly; a UR5 batch on device tensors equals the restatement (the reference's matrices + an exact enumerating QP
    solver) target by target, without and with inequality rows."""
    import torch
    from rtbhip import urdf as U
    from helpers import chain_from_ets
    panda, _ = _panda_limited()
    with pytest.raises(rtbhip.RtbHipError):
        panda.ikine_QP(np.eye(4), kq=0.1, ps=0.3, pi=0.3)
    with pytest.raises(rtbhip.RtbHipError):
        panda.ikine_QP(np.eye(4), kj=0.0)
    with pytest.raises(rtbhip.RtbHipError):
        U.load("px100").ets().ikine_QP(np.eye(4), km=1.0)
    with pytest.raises(rtbhip.RtbHipError):
        U.load("px100").ets().ikine_QP(np.eye(4), kq=1.0)
    e = U.load("UR5").ets()
    e.qlim = np.clip(e.qlim, -np.pi, np.pi)
    ch = chain_from_ets(e)
    rng = np.random.default_rng(66)
    N = 200
    qs = rng.uniform(ch.qlim[0] * 0.8, ch.qlim[1] * 0.8, (N, 6))
    Tep = oracle.fkine(ch, qs)
    q0 = qs + 0.15 * rng.normal(size=qs.shape)
    for kw in (dict(kj=0.01), dict(kj=0.01, kq=1.0, ps=0.0, pi=0.5)):
        sol = e.ikine_QP(torch.from_numpy(Tep).cuda(), q0=torch.from_numpy(q0).cuda(), seed=8, slimit=3, **kw)
        for i in range(0, N, 7):
            rs = np.array([q0[i]] + [e.ik_restart(8, i, d) for d in range(1, 3)])
            o = oracle.ikine_py(ch, Tep[i], rs, step="qp", slimit=3, ks=1.0, **kw)
            assert (o[1], o[2], o[3]) == (sol.each["success"][i], sol.each["iterations"][i], sol.each["searches"][i])
            if o[1]:
                nt.assert_allclose(sol.q[i], o[0], atol=1e-7)
        assert sol.each["success"].mean() > 0.8


def test_branches_of_a_tree_robot_read_the_robot_wide_q():
    """rtbhip_chain_set_q_width: YuMi's two arms evaluated from ONE (N, 18) array -- per-chain kernels, the fleet launch, host
    and device pointers -- equal the path-numbered chains on the picked columns."""
    import torch
    from rtbhip import urdf as U
    y = U.load("YuMi")
    rng = np.random.default_rng(18)
    q = rng.uniform(-1.0, 1.0, (1000, y.n))
    ends = ("gripper_r_finger_r", "gripper_l_finger_l")
    wide = [y.ets(end=e, compact=False) for e in ends]
    local = [y.ets(end=e) for e in ends]
    qd = torch.from_numpy(q).cuda()
    Tf, Jf = rtbhip.fleet_fkine_jacob(wide, [qd, qd])
    for w, l, Tw, Jw in zip(wide, local, Tf, Jf):
        qs = np.ascontiguousarray(q[:, w.jindices])
        Tl, Jl = l.fkine_jacob0(qs)
        nt.assert_array_equal(w.eval(q), Tl)
        nt.assert_array_equal(w.jacob0(q), Jl)
        nt.assert_array_equal(Tw.cpu().numpy(), Tl)
        nt.assert_array_equal(Jw.cpu().numpy(), Jl)
        nt.assert_array_equal(w.hessian0(q[:50]), l.hessian0(qs[:50]))


def test_ik_row_blocks_with_target_base_equal_the_whole_batch():
    """Sharded IK: each row block solved by its own call inside `ShardedBatch.ik_rows()` / `rtbhip.ik_target_base(begin)`
    returns, bit for bit, the rows of the single call over all targets -- C flavour, Python flavour and IK_QP."""
    import torch
    ets, ch = _panda_limited()
    rng = np.random.default_rng(31)
    N = 5000
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    Tep[::37, :3, 3] += 2.5
    Tt = torch.from_numpy(Tep).cuda()
    runs = (lambda T: [x.cpu().numpy() for x in ets.ik_LM(T, seed=4, slimit=20)],
            lambda T: (lambda s: [s.q, s.each["success"], s.each["iterations"], s.each["searches"], s.each["residual"]])(ets.ikine_LM(T, seed=4, slimit=20)),
            lambda T: (lambda s: [s.q, s.each["success"], s.each["iterations"], s.each["searches"], s.each["residual"]])(ets.ikine_QP(T, seed=4, slimit=20, kj=0.01)))
    for run in runs:
        whole = run(Tt)
        assert whole[3].max() > 3
        for world in (2, 3):
            parts = []
            for r in range(world):
                sb = rtbhip.ShardedBatch(N, r, world)
                with sb.ik_rows():
                    parts.append(run(sb.local(Tt).contiguous()))
            for k in range(5):
                nt.assert_array_equal(np.concatenate([p[k] for p in parts]), whole[k])
    plain = runs[0](Tt[N // 2:].contiguous())
    assert not np.array_equal(plain[0], runs[0](Tt)[0][N // 2:])


def test_init_and_shutdown_keep_handles_usable():
    lib = rtbhip.lib()
    assert lib.rtbhip_init(-1) == 0 and lib.rtbhip_init(1) == 0
    assert lib.rtbhip_init(10 ** 6) != 0
    ets = rtbhip.models.Panda().ets()
    q = np.random.default_rng(0).uniform(-3, 3, (100, 7))
    T0 = ets.eval(q)
    arm = rtbhip.models.DH.Puma560()
    t0 = arm.rne(q[:, :6], q[:, :6], q[:, :6])
    lib.rtbhip_shutdown()                                   # drops every cached device table
    nt.assert_array_equal(ets.eval(q), T0)                  # ... which are re-uploaded on next use
    nt.assert_array_equal(arm.rne(q[:, :6], q[:, :6], q[:, :6]), t0)
    assert ets.ik_LM(T0[0], seed=1)[1] in (0, 1)


def test_ik_small_chains_and_errors():
    ET = rtbhip.ET
    arm = ET.Rz() * ET.tx(1.0) * ET.Rz() * ET.tx(1.0)            # planar 2R
    ch = chains.Chain([("Rz",), ("tx", 1.0), ("Rz",), ("tx", 1.0)])
    Tep = oracle.fkine(ch, np.array([[0.3, 0.8], [-1.0, 1.2]]))
    q, ok, it, se, E = arm.ik_LM(Tep, mask=[1, 1, 0, 0, 0, 1], joint_limits=False)
    assert ok.all()
    nt.assert_allclose(oracle.fkine(ch, q)[:, :2, 3], Tep[:, :2, 3], atol=2e-3)
    big = rtbhip.ETS([ET.Rz() for _ in range(33)])             # 33 joints: beyond RTBHIP_MAX_JOINTS, refused when the chain is made
    with pytest.raises(rtbhip.RtbHipError):
        big.ik_LM(np.eye(4))
    perm = ET.Rz(jindex=1) * ET.tx(1.0) * ET.Rz(jindex=0)
    with pytest.raises(rtbhip.RtbHipError):
        perm.ik_LM(np.eye(4))


def test_large_batch_64bit_indexing():
    """N = 20,000,003 Panda configurations (q 1.1 GB, T 2.6 GB, J 6.7 GB on the device): element offsets pass
    2^31, the last tile is ragged; the first and last rows must equal the oracle."""
    import torch
    N = 20000003
    ets = rtbhip.models.Panda().ets()
    ch = chains.panda_ets()
    g = torch.Generator(device=DEV()).manual_seed(3)
    q = (torch.rand((N, 7), dtype=torch.float64, device=DEV(), generator=g) - 0.5) * 6.0
    T, J = ets.fkine_jacob0(q)
    torch.cuda.synchronize()
    idx = torch.cat([torch.arange(0, 70), torch.arange(N - 70, N), torch.tensor([N // 2, 2 ** 24 + 1, 16777216 + 64 * 1000 + 5])]).cuda()
    qh = q[idx].cpu().numpy()
    nt.assert_allclose(T[idx].cpu().numpy(), oracle.fkine(ch, qh), atol=TOL)
    nt.assert_allclose(J[idx].cpu().numpy(), oracle.jacob0(ch, qh), atol=TOL)
    assert bool(torch.isfinite(T[-1]).all()) and float((T[:, 3, 3] - 1).abs().max()) == 0.0
    del T, J
    tau_arm = rtbhip.models.DH.Panda()
    tau = tau_arm.rne(q, q, q)                         # 4 x 1.1 GB more: the RNE tile indexing at the same size
    tab = chains.panda_dh()
    ref = oracle.rne_dh(tab.L24(), 1, qh, qh, qh, -tab.gravity)
    got = tau[idx].cpu().numpy()
    nt.assert_allclose(got, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())

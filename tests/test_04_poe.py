"""Product-of-exponentials robots (reference robot/PoERobot.py, tests/test_PoERobot.py).

Pins, in this order:
  1. the oracle (oracle/poe.py): its closed-form PoE kinematics against its restatement of the reference's own PoE -> ETS lowering
     evaluated by the reference's COMPILED fknm (oracle/_ref) -- the statement of tests/test_PoERobot.py:30-34 and :70-74 for
     the reference's two robots at the reference's q (there to 7 decimals, here to 1e-12);
  2. the product's direct lowering of the twists (csrc/chain.cpp compile_poe) and its `ets()` (the reference's recipe), kernel
     bodies replayed on the CPU by tests/emu, against the oracle's closed form;
  3. (-m gpu) the same through the C ABI on the device, 50 random twist chains, ragged batch sizes, Hessian and IK on a PoE chain.
"""
import ctypes as C

import numpy as np
import numpy.testing as nt
import pytest

from oracle import oracle, poe, ref_harness

TOL = 1e-10                       # north-star pose tolerance; observed ~1e-15


def product_robot(r, **kw):
    import rtbhip
    links = []
    for S in r.S:
        if np.linalg.norm(S[3:]) == 0.0:
            links.append(rtbhip.PoEPrismatic(S[:3]))
        else:
            w, v = S[3:], S[:3]
            links.append(rtbhip.PoERevolute(w, np.cross(w, v)))          # a point of the axis: w x v
    return rtbhip.PoERobot(links, r.T0, **kw)


def random_poe(rng, n):
    S = []
    for _ in range(n):
        axis = rng.normal(size=3)
        if rng.random() < 0.25:
            axis = np.eye(3)[rng.integers(3)] * rng.choice([-1.0, 1.0])       # coordinate axes: the degenerate directions
        if rng.random() < 0.3:
            S.append(poe.unit_prismatic(axis))
        else:
            S.append(poe.unit_revolute(axis, rng.uniform(-0.5, 0.5, 3) * (rng.random() > 0.2)))
    T0 = np.eye(4)
    T0[:3, :3] = poe.rodrigues(*(lambda a: (a / np.linalg.norm(a), rng.uniform(-3, 3)))(rng.normal(size=3)))
    T0[:3, 3] = rng.uniform(-0.5, 0.5, 3)
    return poe.PoE(S, T0)


# ------------------------------------------------------------------------------------------------ 1. the oracle itself
@pytest.mark.parametrize("make", [poe.test_robot_2rpr, poe.test_robot_3rp])
def test_oracle_closed_form_equals_reference_lowering_on_compiled_fknm(make):
    if not ref_harness.available():
        pytest.skip("oracle/_ref not built")
    r, q = make()
    ref = ref_harness.RefETS(r.chain())
    nt.assert_allclose(r.fkine(q), ref.fkine(q), atol=1e-12)            # tests/test_PoERobot.py:30, :70
    nt.assert_allclose(r.jacob0(q), ref.jacob0(q), atol=1e-12)          # :33, :73
    nt.assert_allclose(r.jacobe(q), ref.jacobe(q), atol=1e-12)          # :34, :74
    Q = np.random.default_rng(3).uniform(-3, 3, (64, r.n))
    nt.assert_allclose(r.fkine(Q), ref.fkine(Q), atol=1e-12)
    nt.assert_allclose(r.jacob0(Q), ref.jacob0_batch(Q), atol=1e-12)
    nt.assert_allclose(r.jacobe(Q), ref.jacobe_batch(Q), atol=1e-12)


def test_oracle_twist_exponential_properties():
    rng = np.random.default_rng(0)
    for _ in range(20):
        S = poe.unit_revolute(rng.normal(size=3), rng.normal(size=3))
        a, b = rng.uniform(-3, 3, 2)
        nt.assert_allclose(poe.twist_exp(S, a) @ poe.twist_exp(S, b), poe.twist_exp(S, a + b), atol=1e-13)
        T = poe.twist_exp(S, a)
        nt.assert_allclose(T[:3, :3] @ T[:3, :3].T, np.eye(3), atol=1e-14)
        p = np.cross(S[3:], S[:3])                                       # a point on the axis does not move
        nt.assert_allclose(T[:3, :3] @ p + T[:3, 3], p, atol=1e-14)
    P = poe.unit_prismatic([1.0, 2.0, -2.0])
    nt.assert_allclose(poe.twist_exp(P, 0.6)[:3, 3], 0.6 * np.array([1, 2, -2]) / 3.0, atol=1e-15)


# ------------------------------------------------------------------------------------------------ 2. product lowering on the CPU replay
@pytest.mark.parametrize("make", [poe.test_robot_2rpr, poe.test_robot_3rp])
def test_emu_reference_robots(make):
    import emu_harness as emu
    import rtbhip
    r, q = make()
    robot = product_robot(r)
    assert robot.n == 4 and len(robot.links) == 6
    Q = np.vstack([q, np.random.default_rng(5).uniform(-3, 3, (20, 4))])
    for chain in (robot._path(None, None), robot.ets(), rtbhip.ERobot(robot.ets()).ets()):   # twists; the reference's recipe; Robot(r.ets())
        for reg in (True, False):
            T, J0, _ = emu.kin(chain, Q, reg=reg)
            _, Je, _ = emu.kin(chain, Q, frame=1, reg=reg)
            nt.assert_allclose(T, r.fkine(Q), atol=1e-12)
            nt.assert_allclose(J0, r.jacob0(Q), atol=1e-12)
            nt.assert_allclose(Je, r.jacobe(Q), atol=1e-12)
    # the ET list is the reference recipe's (same elements in the same order, same constants)
    want = r.lowered()
    got = [(e.axis,) if e.isjoint else (e.axis, e.eta) for e in robot.ets()]
    assert [g[0] for g in got] == [w[0] for w in want]
    for g, w in zip(got, want):
        if len(w) > 1:
            assert abs(g[1] - w[1]) < 1e-12
    assert list(robot.ets().jindices) == [0, 1, 2, 3]


def test_emu_random_twist_chains():
    import emu_harness as emu
    rng = np.random.default_rng(11)
    for k in range(50):
        n = 1 + k % 10
        r = random_poe(rng, n)
        robot = product_robot(r)
        Q = rng.uniform(-3, 3, (4, n))
        reg = n <= 8
        T, J0, _ = emu.kin(robot._path(None, None), Q, reg=reg)
        _, Je, _ = emu.kin(robot._path(None, None), Q, frame=1, reg=reg)
        nt.assert_allclose(T, r.fkine(Q), atol=1e-12)
        nt.assert_allclose(J0, r.jacob0(Q), atol=1e-12)
        nt.assert_allclose(Je, r.jacobe(Q), atol=1e-12)
        # the reference's roll-pitch-yaw re-expression drops elements below 1e-8 (np.isclose): equal to that level only
        T2, J2, _ = emu.kin(robot.ets(), Q, reg=reg)
        nt.assert_allclose(T2, T, atol=1e-7)
        nt.assert_allclose(J2, J0, atol=1e-7)


def test_poe_abi_validation():
    import rtbhip
    lib = rtbhip.lib()
    h = C.c_uint64(0)
    ok = np.ascontiguousarray([poe.unit_revolute([0, 0, 1], [0.1, 0, 0]), poe.unit_prismatic([0, 1, 0])])
    T0 = np.eye(4)
    assert lib.rtbhip_chain_create_poe(ok.ctypes.data_as(C.c_void_p), 2, T0.ctypes.data_as(C.c_void_p), None, C.byref(h)) == 0
    n, m, qw = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.rtbhip_chain_info(h.value, C.byref(n), C.byref(m), C.byref(qw)) == 0
    assert (n.value, qw.value) == (2, 2) and m.value == 5           # C J C J C
    assert lib.rtbhip_chain_destroy(h.value) == 0
    assert lib.rtbhip_chain_create_poe(ok.ctypes.data_as(C.c_void_p), 2, None, None, C.byref(h)) == 0       # T0 NULL = identity
    assert lib.rtbhip_chain_destroy(h.value) == 0
    bad = ok.copy(); bad[0, 3:] *= 2.0                               # |w| = 2
    assert lib.rtbhip_chain_create_poe(bad.ctypes.data_as(C.c_void_p), 2, None, None, C.byref(h)) == -1
    assert b"unit revolute" in lib.rtbhip_last_error()
    bad = ok.copy(); bad[0, :3] += 0.3 * bad[0, 3:]                  # a pitch
    assert lib.rtbhip_chain_create_poe(bad.ctypes.data_as(C.c_void_p), 2, None, None, C.byref(h)) == -1
    assert b"pitch" in lib.rtbhip_last_error()
    bad = ok.copy(); bad[1, :3] *= 0.5                               # prismatic, |v| = 0.5
    assert lib.rtbhip_chain_create_poe(bad.ctypes.data_as(C.c_void_p), 2, None, None, C.byref(h)) == -1
    bad = ok.copy(); bad[1, 0] = np.nan
    assert lib.rtbhip_chain_create_poe(bad.ctypes.data_as(C.c_void_p), 2, None, None, C.byref(h)) == -1
    T0b = np.eye(4); T0b[3, 0] = 1.0
    assert lib.rtbhip_chain_create_poe(ok.ctypes.data_as(C.c_void_p), 2, T0b.ctypes.data_as(C.c_void_p), None, C.byref(h)) == -1
    assert lib.rtbhip_chain_create_poe(None, 2, None, None, C.byref(h)) == -1
    assert lib.rtbhip_chain_create_poe(ok.ctypes.data_as(C.c_void_p), 2, None, None, None) == -1
    too_many = np.tile(ok[0], (33, 1))
    assert lib.rtbhip_chain_create_poe(too_many.ctypes.data_as(C.c_void_p), 33, None, None, C.byref(h)) == -3


def test_poe_robot_surface():
    import rtbhip
    r, q = poe.test_robot_2rpr()
    links = [rtbhip.PoERevolute([0, 0, 1], [0, 0, 0], name="foo"), rtbhip.PoERevolute([0, 1, 0], [0, 0, 0.2]),
             rtbhip.PoEPrismatic([0, 1, 0]), rtbhip.PoERevolute([0, -1, 0], [0.2, 0, 0.5])]
    robot = rtbhip.PoERobot(links, r.T0, name="2RPR")
    assert len(links) == 6                                            # the reference inserts its base / ee links into the caller's list
    nt.assert_allclose(robot.twists, r.S, atol=1e-15)
    assert robot.nbranches() == 0 and "PoERevolute" in repr(robot) and "T0" in str(robot)
    assert [l.isrevolute for l in robot][1:5] == [True, True, False, True]
    nt.assert_allclose(poe.twist_exp(robot.links[-1].S, 1.0), r.T0, atol=1e-12)      # the ee link's twist is log(T0)
    with pytest.raises(TypeError):
        rtbhip.PoERobot([rtbhip.PoELink(np.zeros(6))], np.eye(4))
    with pytest.raises(TypeError):
        robot.fkine(q, bogus=1)                                       # unknown keywords are refused, never ignored
    with pytest.raises(ValueError):
        rtbhip.PoERevolute([0, 0, 0], [0, 0, 0])
    e = robot.ets(1, 2)                                               # a sub-path in the elementary form
    assert e.n == 2
    # a revolute joint about the base x axis through the origin: the reference's frame recipe degenerates (NaN); ours does not
    rx = rtbhip.PoERobot([rtbhip.PoERevolute([1, 0, 0], [0, 0, 0])], np.eye(4))
    assert all(np.isfinite(x.eta) for x in rx.ets() if not x.isjoint)


# ------------------------------------------------------------------------------------------------ 3. the device
@pytest.mark.gpu
@pytest.mark.parametrize("make", [poe.test_robot_2rpr, poe.test_robot_3rp])
def test_gpu_reference_robots(make):
    import rtbhip
    r, q = make()
    robot = product_robot(r)
    as_ets = rtbhip.ERobot(robot.ets())                               # tests/test_PoERobot.py:27, :68  Robot(r.ets())
    nt.assert_allclose(robot.fkine(q), r.fkine(q), atol=TOL)
    nt.assert_allclose(robot.jacob0(q), r.jacob0(q), atol=TOL)
    nt.assert_allclose(robot.jacobe(q), r.jacobe(q), atol=TOL)
    nt.assert_allclose(robot.fkine(q), as_ets.fkine(q), atol=TOL)     # :30 / :70
    nt.assert_allclose(robot.jacob0(q), as_ets.jacob0(q), atol=TOL)   # :33 / :73
    nt.assert_allclose(robot.jacobe(q), as_ets.jacobe(q), atol=TOL)   # :34 / :74
    for N in (0, 1, 63, 64, 65, 1000, 4097):
        Q = np.random.default_rng(N).uniform(-3, 3, (N, 4))
        if N == 0:
            assert robot.fkine(Q).shape == (0, 4, 4)
            continue
        if N == 1:
            Q = Q.reshape(1, 4)
        T, J0, Je = robot.fkine(Q), robot.jacob0(Q), robot.jacobe(Q)
        nt.assert_allclose(np.asarray(T).reshape(-1, 4, 4), r.fkine(Q), atol=TOL)
        nt.assert_allclose(np.asarray(J0).reshape(-1, 6, 4), r.jacob0(Q), atol=TOL)
        nt.assert_allclose(np.asarray(Je).reshape(-1, 6, 4), r.jacobe(Q), atol=TOL)


@pytest.mark.gpu
def test_gpu_random_twist_chains():
    import torch
    rng = np.random.default_rng(11)
    worst = 0.0
    for k in range(50):
        n = 1 + k % 12
        r = random_poe(rng, n)
        robot = product_robot(r)
        Q = rng.uniform(-3, 3, (130, n))
        Qd = torch.from_numpy(Q).cuda()
        T, J0 = robot._path(None, None).fkine_jacob0(Qd)              # the fused headline op, device pointers
        Je = robot.jacobe(Qd)
        eT = np.abs(T.cpu().numpy() - r.fkine(Q)).max()
        eJ = np.abs(J0.cpu().numpy() - r.jacob0(Q)).max()
        eE = np.abs(Je.cpu().numpy() - r.jacobe(Q)).max()
        worst = max(worst, eT, eJ, eE)
        assert max(eT, eJ, eE) <= TOL, (k, n, eT, eJ, eE)
        nt.assert_allclose(robot.fkine(Q), r.fkine(Q), atol=TOL)      # host-pointer path
    assert worst < 1e-12


@pytest.mark.gpu
def test_gpu_poe_chain_serves_the_rest_of_the_path():
    """A PoE handle is a chain like any other: Hessian against finite differences of the closed-form Jacobian, LM inverse
    kinematics landing on the closed-form pose, fkine_all-style frames are not needed."""
    import rtbhip
    r, q = poe.test_robot_3rp()
    robot = product_robot(r)
    H = robot.hessian0(q)
    h = 1e-6
    for j in range(4):
        dq = np.zeros(4); dq[j] = h
        fd = (r.jacob0(q + dq) - r.jacob0(q - dq)) / (2 * h)
        nt.assert_allclose(H[j], fd, atol=1e-7)
    rng = np.random.default_rng(2)
    r6 = random_poe(rng, 6)
    robot6 = product_robot(r6)
    robot6.qlim = np.array([[-3.0] * 6, [3.0] * 6])
    qs = rng.uniform(-1.5, 1.5, (200, 6))
    Tep = r6.fkine(qs)
    qsol, ok, it, se, E = robot6.ik_LM(Tep, seed=4)
    assert ok.mean() > 0.95
    good = ok.astype(bool)
    assert E[good].max() < 1e-6
    for T, Tt in zip(r6.fkine(qsol[good]), Tep[good]):                # the reference's own criterion (tests/test_IK.py:15): E of the CLOSED-FORM pose
        e = oracle.angle_axis(T, Tt)
        assert 0.5 * e @ e < 1e-5
    N = 200000
    Q = rng.uniform(-np.pi, np.pi, (N, 6))
    T = robot6.fkine(Q)
    R = T[:, :3, :3]
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-12
    idx = rng.integers(0, N, 64)
    nt.assert_allclose(T[idx], r6.fkine(Q[idx]), atol=TOL)

"""Dynamics-mixin terms: inertia / coriolis / gravload / itorque / accel as fused kernels.
   * the oracle's restatement of robot/Dynamics.py (oracle/oracle.py *_dh) is pinned on the reference's
     own goldens (tests/test_DHRobot.py:1092-1202, lifted into tests/golden/reference_literals.json);
   * the kernels' per-lane body (dyn_device.h) runs on the CPU through tests/emu against that oracle;
   * the GPU kernels run through the C ABI against the oracle and the goldens."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from oracle import oracle, chains
from helpers import literals

LIT = literals()
QN = np.array([0, np.pi / 4, np.pi, 0, np.pi / 4, 0])


def _puma():
    t = chains.puma560()
    return t, t.L24(), -t.gravity


def test_oracle_restatement_matches_reference_goldens():
    t, L, gc = _puma()
    nt.assert_array_almost_equal(oracle.inertia_dh(L, 0, QN)[0], LIT["D_puma_inertia"], decimal=4)
    nt.assert_array_almost_equal(oracle.coriolis_dh(L, 0, QN, LIT["D_puma_coriolis_qd"])[0], LIT["D_puma_coriolis"], decimal=4)
    nt.assert_array_almost_equal(oracle.accel_dh(L, 0, QN, LIT["D_puma_accel_qd"], LIT["D_puma_accel_torque"], gc)[0],
                                 LIT["D_puma_accel"], decimal=4)
    z = np.zeros(6)
    nt.assert_array_almost_equal(oracle.rne_dh(L, 0, QN, z, z, gc)[0], LIT["D_puma_gravload"], decimal=4)
    nt.assert_array_almost_equal(oracle.rne_dh(L, 0, QN, z, LIT["D_puma_itorque_qdd"], [0, 0, 0])[0],
                                 LIT["D_puma_itorque"], decimal=4)


@pytest.mark.parametrize("robot", ["puma", "panda"])
def test_emu_kernel_body_vs_oracle(robot):
    import emu_harness as emu
    t = chains.puma560() if robot == "puma" else chains.panda_dh()
    mdh = 0 if robot == "puma" else 1
    L, gc, n = t.L24(), -t.gravity, t.L24().shape[0]
    rng = np.random.default_rng(8)
    q = rng.uniform(t.qlim[:, 0], t.qlim[:, 1], (9, n))
    qd, tq = rng.normal(size=(9, n)), rng.normal(size=(9, n)) * 5
    M = emu.dyn(L, mdh, 0, q)
    nt.assert_allclose(M, oracle.inertia_dh(L, mdh, q), rtol=1e-11, atol=1e-12)
    nt.assert_allclose(M, np.swapaxes(M, 1, 2), atol=1e-12)
    Cm = emu.dyn(L, mdh, 1, q, qd)
    nt.assert_allclose(Cm, oracle.coriolis_dh(L, mdh, q, qd), rtol=1e-10, atol=1e-11)
    a = emu.dyn(L, mdh, 2, q, qd, tq, gc)
    ref = oracle.accel_dh(L, mdh, q, qd, tq, gc)
    nt.assert_allclose(a, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())


def test_emu_goldens_puma_qn():
    import emu_harness as emu
    t, L, gc = _puma()
    nt.assert_array_almost_equal(emu.dyn(L, 0, 0, QN)[0], LIT["D_puma_inertia"], decimal=4)
    nt.assert_array_almost_equal(emu.dyn(L, 0, 1, QN, LIT["D_puma_coriolis_qd"])[0], LIT["D_puma_coriolis"], decimal=4)
    nt.assert_array_almost_equal(emu.dyn(L, 0, 2, QN, LIT["D_puma_accel_qd"], LIT["D_puma_accel_torque"], gc)[0],
                                 LIT["D_puma_accel"], decimal=4)


@pytest.mark.gpu
def test_gpu_goldens_and_shapes_puma():
    """reference tests/test_DHRobot.py:1092-1202: single q -> (n,) / (n,n); stacked q -> leading batch axis."""
    puma = rtbhip.models.DH.Puma560()
    q2 = np.c_[QN, QN].T
    nt.assert_array_almost_equal(puma.inertia(QN), LIT["D_puma_inertia"], decimal=4)
    qd = LIT["D_puma_coriolis_qd"]
    nt.assert_array_almost_equal(puma.coriolis(QN, qd), LIT["D_puma_coriolis"], decimal=4)
    C1 = puma.coriolis(q2, np.c_[qd, qd].T)
    assert C1.shape == (2, 6, 6)
    nt.assert_array_almost_equal(C1[1], LIT["D_puma_coriolis"], decimal=4)
    nt.assert_array_almost_equal(puma.gravload(QN), LIT["D_puma_gravload"], decimal=4)
    nt.assert_array_almost_equal(puma.gravload(q2)[1], LIT["D_puma_gravload"], decimal=4)
    qdd = LIT["D_puma_itorque_qdd"]
    nt.assert_array_almost_equal(puma.itorque(QN, qdd), LIT["D_puma_itorque"], decimal=4)
    nt.assert_array_almost_equal(puma.itorque(q2, np.c_[qdd, qdd].T)[0], LIT["D_puma_itorque"], decimal=4)
    aq, at = LIT["D_puma_accel_qd"], LIT["D_puma_accel_torque"]
    nt.assert_array_almost_equal(puma.accel(QN, aq, at), LIT["D_puma_accel"], decimal=4)
    nt.assert_array_almost_equal(puma.accel(q2, np.c_[aq, aq].T, np.c_[at, at].T)[1], LIT["D_puma_accel"], decimal=4)


@pytest.mark.parametrize("robot", ["puma", "panda"])
def test_emu_rne_at_rest_variant(robot):
    """qd = NULL on an all-revolute chain takes the acceleration-only forward recursion with gravity as the base's acceleration
    (k_rne_atrest): gravload (qdd NULL too), itorque (no gravity), and the general qd = NULL call with gravity, qdd and an
    external wrench must equal the full recursion fed zeros."""
    import emu_harness as emu
    t = chains.puma560() if robot == "puma" else chains.panda_dh()
    mdh, L, gc, n = (0 if robot == "puma" else 1), t.L24(), -t.gravity, t.L24().shape[0]
    rng = np.random.default_rng(12)
    q = rng.uniform(t.qlim[:, 0], t.qlim[:, 1], (70, n))
    qdd = rng.normal(size=(70, n))
    z = np.zeros_like(q)
    fext = np.array([1.0, -2.0, 0.5, 0.1, 0.2, -0.3])
    for args in ((None, gc, None), (qdd, [0, 0, 0], None), (qdd, gc, fext)):
        acc, g, f = args
        fast = emu.rne(L, mdh, q, None, acc, g, fext=f)
        full = emu.rne(L, mdh, q, z, z if acc is None else acc, g, fext=f)
        ref = oracle.rne_dh(L, mdh, q, z, z if acc is None else acc, g, fext=f)
        nt.assert_allclose(fast, ref, rtol=1e-12, atol=1e-12)
        nt.assert_allclose(fast, full, rtol=1e-14, atol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["puma", "panda"])
def test_gpu_rne_at_rest_variant(robot):
    """rtbhip_rne with qd = NULL (k_rne_atrest) on the device: gravload, itorque and a call with gravity, qdd and an external wrench
    against the oracle's full recursion fed zeros; host and device pointers; ragged tile."""
    import torch
    rob = rtbhip.models.DH.Puma560() if robot == "puma" else rtbhip.models.DH.Panda()
    t = chains.puma560() if robot == "puma" else chains.panda_dh()
    mdh, L, gc, n = rob.mdh, t.L24(), -t.gravity, rob.n
    rng = np.random.default_rng(13)
    N = 1000 + 37
    q = rng.uniform(t.qlim[:, 0], t.qlim[:, 1], (N, n))
    qdd = rng.normal(size=(N, n))
    z = np.zeros_like(q)
    fext = np.array([1.0, -2.0, 0.5, 0.1, 0.2, -0.3])
    nt.assert_allclose(rob.gravload(q), oracle.rne_dh(L, mdh, q, z, z, gc), rtol=1e-12, atol=1e-12)
    nt.assert_allclose(rob.itorque(q, qdd), oracle.rne_dh(L, mdh, q, z, qdd, [0, 0, 0]), rtol=1e-12, atol=1e-12)
    ref = oracle.rne_dh(L, mdh, q, z, qdd, gc, fext=fext)
    host = rob.rne(q, None, qdd, fext=fext)
    nt.assert_allclose(host, ref, rtol=1e-12, atol=1e-12)
    dev = rob.rne(torch.from_numpy(q).cuda(), None, torch.from_numpy(qdd).cuda(), fext=fext)
    nt.assert_array_equal(dev.cpu().numpy(), host)


def _coriolis_scale_cases(n, rng):
    """qd rows of very different scales (and mixed within a row, zero, one nonzero entry).  dyn_device.h evaluates column k as the bilinear form
    B(qd, e_k) in one two-field pass -- linear in qd, so its accuracy cannot depend on the scale or on the spread within a row.  (Rounds 1-3
    probed with qd +- s e_k and sent rows spanning more than 2^16 to the reference's own 28-pass scheme; the tolerances below are from then:
    for row 6 -- nine orders of magnitude -- 1e-12 of max|C| is the rounding of the REFERENCE's combination, which the oracle restates.)"""
    base = rng.normal(size=(8, n))
    rows = [base[0] * 1e-9, base[1] * 1e-3, base[2], base[3] * 1e3, base[4] * 1e9, np.zeros(n), base[6] * np.logspace(-6, 3, n)]
    one = np.zeros(n); one[n // 2] = -3.7
    rows.append(one)
    rows.append(base[7] * np.logspace(-1.5, 1.5, n))           # a moderate spread
    return np.array(rows)


@pytest.mark.parametrize("robot", ["puma", "panda"])
def test_emu_coriolis_accuracy_is_scale_free(robot):
    import emu_harness as emu
    t = chains.puma560() if robot == "puma" else chains.panda_dh()
    mdh, L, n = (0 if robot == "puma" else 1), t.L24(), t.L24().shape[0]
    rng = np.random.default_rng(77)
    qd = _coriolis_scale_cases(n, rng)
    q = rng.uniform(t.qlim[:, 0], t.qlim[:, 1], (len(qd), n))
    C = emu.dyn(L, mdh, 1, q, qd)
    ref = oracle.coriolis_dh(L, mdh, q, qd)
    for i in range(len(qd)):
        scale = np.abs(ref[i]).max()
        assert np.abs(C[i] - ref[i]).max() <= (1e-12 if i in (6, 8) else 1e-13) * scale, (i, np.abs(C[i] - ref[i]).max(), scale)
    assert np.all(C[5] == 0.0)


@pytest.mark.gpu
def test_gpu_coriolis_accuracy_is_scale_free():
    rob, t = rtbhip.models.DH.Panda(), chains.panda_dh()
    rng = np.random.default_rng(78)
    qd = _coriolis_scale_cases(7, rng)
    q = rng.uniform(t.qlim[:, 0], t.qlim[:, 1], (len(qd), 7))
    C = rob.coriolis(q, qd)
    ref = oracle.coriolis_dh(t.L24(), 1, q, qd)
    for i in range(len(qd)):
        assert np.abs(C[i] - ref[i]).max() <= (1e-12 if i in (6, 8) else 1e-13) * np.abs(ref[i]).max()
    assert np.all(C[5] == 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("robot,N", [("puma", 1000), ("panda", 4097), ("panda", 63)])
def test_gpu_vs_oracle_and_identities(robot, N):
    import torch
    rob = rtbhip.models.DH.Puma560() if robot == "puma" else rtbhip.models.DH.Panda()
    t = chains.puma560() if robot == "puma" else chains.panda_dh()
    mdh, L, gc, n = rob.mdh, t.L24(), -t.gravity, rob.n
    rng = np.random.default_rng(N)
    q = rng.uniform(t.qlim[:, 0], t.qlim[:, 1], (N, n))
    qd, tq, qdd = rng.normal(size=(N, n)), rng.normal(size=(N, n)) * 5, rng.normal(size=(N, n))
    k = min(N, 200)                                    # the oracle composition is a Python loop: bounded sample
    M = rob.inertia(q)
    nt.assert_allclose(M[:k], oracle.inertia_dh(L, mdh, q[:k]), rtol=1e-11, atol=1e-12)
    Cm = rob.coriolis(q, qd)
    nt.assert_allclose(Cm[:k], oracle.coriolis_dh(L, mdh, q[:k], qd[:k]), rtol=1e-10, atol=1e-11)
    a = rob.accel(q, qd, tq)
    ref = oracle.accel_dh(L, mdh, q[:k], qd[:k], tq[:k], gc)
    nt.assert_allclose(a[:k], ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    # size-independent identities over the whole batch
    nt.assert_allclose(M, np.swapaxes(M, 1, 2), atol=1e-11)                       # symmetric
    assert np.linalg.eigvalsh(M).min() > 0                                        # positive definite
    nt.assert_allclose(rob.itorque(q, qdd), np.einsum("nij,nj->ni", M, qdd), rtol=1e-10, atol=1e-10)
    full = rob.rne(q, np.zeros_like(q), np.zeros_like(q))                         # NULL qd = the at-rest recursion, zeros = the full one
    nt.assert_allclose(rob.gravload(q), full, rtol=0, atol=1e-14 * np.abs(full).max())
    # forward dynamics inverts inverse dynamics: rne(q, qd, accel(q, qd, tau)) == tau
    back = rob.rne(q, qd, a)
    nt.assert_allclose(back, tq, rtol=1e-8, atol=1e-8 * np.abs(tq).max())
    # device-pointer path == host-pointer path
    qt, qdt, tqt = (torch.from_numpy(x).cuda() for x in (q, qd, tq))
    nt.assert_array_equal(rob.inertia(qt).cpu().numpy(), M)
    nt.assert_array_equal(rob.coriolis(qt, qdt).cpu().numpy(), Cm)
    nt.assert_array_equal(rob.accel(qt, qdt, tqt).cpu().numpy(), a)


def _long_arm(n):
    """An n-joint standard-DH arm (n up to 16): the ten-joint recipe continued, prismatic joints at 4 and 11."""
    rng = np.random.default_rng(77)
    links = []
    for k in range(n):
        Ifull = list(rng.uniform(0.01, 0.1, 3)) + list(rng.uniform(-0.01, 0.01, 3))
        kw = dict(a=0.05 + 0.01 * k, alpha=[0.0, np.pi / 2, -np.pi / 2][k % 3], m=1.0 + 0.1 * k,
                  r=[0.0, 0.0, 0.0] if k % 3 == 0 else list(rng.uniform(-0.05, 0.05, 3)), I=Ifull, Jm=1e-4 * k, G=1.0 + k, B=1e-3, Tc=[0.01, -0.02])
        links.append(rtbhip.PrismaticDH(theta=0.3, qlim=[0.0, 0.4], **kw) if k in (4, 11) else rtbhip.RevoluteDH(d=0.05, **kw))
    return rtbhip.DHRobot(links, name="long%d" % n)


@pytest.mark.parametrize("n", [11, 14, 16])
def test_emu_eleven_to_sixteen_joints(n):
    """inertia / coriolis / accel beyond 10 joints (the spilling instantiations): kernel body against the oracle."""
    import emu_harness as emu
    rob = _long_arm(n)
    L = rob.L24()
    rng = np.random.default_rng(n)
    q, qd, tq = rng.uniform(-1, 1, (3, n)), rng.normal(size=(3, n)), rng.normal(size=(3, n))
    for k in (4, 11):
        if k < n:
            q[:, k] = rng.uniform(0, 0.4, 3)
    gc = -np.array([0.0, 0.0, -9.81])
    nt.assert_allclose(emu.dyn(L, 0, 0, q), oracle.inertia_dh(L, 0, q), rtol=1e-10, atol=1e-11)
    nt.assert_allclose(emu.dyn(L, 0, 1, q, qd), oracle.coriolis_dh(L, 0, q, qd), rtol=1e-9, atol=1e-10)
    ref = oracle.accel_dh(L, 0, q, qd, tq, gc)
    nt.assert_allclose(emu.dyn(L, 0, 2, q, qd, tq, grav_c=gc), ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())


def _mdh_arm_with_a_prismatic_first_joint(n=5):
    rng = np.random.default_rng(5)
    links = []
    for k in range(n):
        kw = dict(a=0.05 + 0.02 * k, alpha=[0.0, np.pi / 2, -np.pi / 2, 0.3][k % 4], m=1.0 + 0.1 * k, r=list(rng.uniform(-0.05, 0.05, 3)),
                  I=np.diag(rng.uniform(0.01, 0.1, 3)), Jm=1e-4 * k, G=1.0 + k)
        links.append(rtbhip.PrismaticMDH(theta=0.3, qlim=[0.0, 0.4], **kw) if k in (0, 3) else rtbhip.RevoluteMDH(d=0.1, **kw))
    return rtbhip.DHRobot(links, name="mdh-p")


def test_reference_inertia_of_an_mdh_chain_with_a_prismatic_first_joint_is_not_symmetric():
    """core/ne.c:187-196 gives link 1 of such a chain the joint rate and acceleration as ANGULAR quantities: the matrix Dynamics.inertia returns
    (and Dynamics.accel solves with, numpy.linalg.solve) is not symmetric.  Pinned on the compiled frne; the oracle restates it; the kernel
    body keeps the full matrix for modified-DH chains with prismatic joints and solves it as it stands (ldl.h lu_solve_mem)."""
    import emu_harness as emu
    from oracle import ref_harness
    rob = _mdh_arm_with_a_prismatic_first_joint()
    L, n = rob.L24(), rob.n
    rng = np.random.default_rng(1)
    q, qd, tq = rng.uniform(-1, 1, (3, n)), rng.normal(size=(3, n)), rng.normal(size=(3, n))
    q[:, 0] = rng.uniform(0, 0.4, 3); q[:, 3] = rng.uniform(0, 0.4, 3)
    Mo = oracle.inertia_dh(L, 1, q)
    assert np.abs(Mo - np.swapaxes(Mo, 1, 2)).max() > 1e-3 * np.abs(Mo).max()
    if ref_harness.available():
        ref = ref_harness.RefRNE(L, 1)
        M = np.array(ref.rne(np.tile(q[0], (n, 1)), np.zeros((n, n)), np.eye(n), gravity=[0, 0, 0]))
        nt.assert_allclose(Mo[0], M, rtol=0, atol=1e-13 * np.abs(M).max())
    gc = -np.array([0.0, 0.0, -9.81])
    nt.assert_allclose(emu.dyn(L, 1, 0, q), Mo, rtol=1e-11, atol=1e-12)
    ref = oracle.accel_dh(L, 1, q, qd, tq, gc)
    nt.assert_allclose(emu.dyn(L, 1, 2, q, qd, tq, grav_c=gc), ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    # the same chain with a revolute first joint: symmetric, the LDL^T path over the full tile
    rob2 = rtbhip.DHRobot([rtbhip.RevoluteMDH(d=0.1, a=0.05, alpha=0.0, m=1.0, r=[0.01, 0.02, 0.03], I=np.diag([0.02, 0.03, 0.04]))] + list(rob.links[1:]))
    L2 = rob2.L24()
    M2 = oracle.inertia_dh(L2, 1, q)
    assert np.abs(M2 - np.swapaxes(M2, 1, 2)).max() < 1e-13 * np.abs(M2).max()
    ref2 = oracle.accel_dh(L2, 1, q, qd, tq, gc)
    nt.assert_allclose(emu.dyn(L2, 1, 2, q, qd, tq, grav_c=gc), ref2, rtol=1e-9, atol=1e-9 * np.abs(ref2).max())


@pytest.mark.gpu
def test_gpu_accel_of_an_mdh_chain_with_a_prismatic_first_joint():
    rob = _mdh_arm_with_a_prismatic_first_joint()
    L, n = rob.L24(), rob.n
    rng = np.random.default_rng(2)
    q, qd, tq = rng.uniform(-1, 1, (70, n)), rng.normal(size=(70, n)), rng.normal(size=(70, n))
    q[:, 0] = rng.uniform(0, 0.4, 70); q[:, 3] = rng.uniform(0, 0.4, 70)
    k = slice(62, 68)
    nt.assert_allclose(rob.inertia(q)[k], oracle.inertia_dh(L, 1, q[k]), rtol=1e-11, atol=1e-12)
    ref = oracle.accel_dh(L, 1, q[k], qd[k], tq[k], rob._gravity_c(None))
    a = rob.accel(q, qd, tq)
    nt.assert_allclose(a[k], ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    # (no forward-then-inverse identity here: the passes return the ROWS of the matrix, superposition would need its transpose, and the
    # reference solves with the matrix as returned -- for this one chain shape rne(q, qd, accel(q, qd, tau)) != tau in the reference as well)


def _ten_joint_arm():
    """A 10-joint DH arm with a prismatic joint, centre-of-mass offsets, full and diagonal inertia tensors, motor inertia
    and friction: the 9- and 10-joint instantiations (one wave per SIMD)."""
    rng = np.random.default_rng(10)
    links = []
    for k in range(10):
        I = rng.uniform(0.01, 0.1, 3)
        Ifull = np.diag(I) if k % 2 else np.diag(I) + 0.002 * (np.ones((3, 3)) - np.eye(3))
        kw = dict(a=0.05 + 0.02 * k, alpha=[0.0, np.pi / 2, -np.pi / 2][k % 3], m=1.0 + 0.1 * k,
                  r=[0.0, 0.0, 0.0] if k % 3 == 0 else list(rng.uniform(-0.05, 0.05, 3)), I=Ifull, Jm=1e-4 * k, G=1.0 + k, B=1e-3, Tc=[0.01, -0.02])
        links.append(rtbhip.PrismaticDH(theta=0.3, qlim=[0.0, 0.4], **kw) if k == 4 else rtbhip.RevoluteDH(d=0.1, **kw))
    return rtbhip.DHRobot(links, name="ten")


@pytest.mark.parametrize("n", [5, 9, 10])
def test_emu_nine_and_ten_joints(n):
    import emu_harness as emu
    rob = _ten_joint_arm()
    rob = rtbhip.DHRobot(rob.links[:n])
    L = rob.L24()
    rng = np.random.default_rng(n)
    q, qd, tq = rng.uniform(-1, 1, (3, n)), rng.normal(size=(3, n)), rng.normal(size=(3, n))
    q[:, 4] = rng.uniform(0, 0.4, 3)
    gc = -np.array([0.0, 0.0, -9.81])
    nt.assert_allclose(emu.dyn(L, 0, 0, q), oracle.inertia_dh(L, 0, q), rtol=1e-11, atol=1e-12)
    nt.assert_allclose(emu.dyn(L, 0, 1, q, qd), oracle.coriolis_dh(L, 0, q, qd), rtol=1e-10, atol=1e-11)
    ref = oracle.accel_dh(L, 0, q, qd, tq, gc)
    nt.assert_allclose(emu.dyn(L, 0, 2, q, qd, tq, grav_c=gc), ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())


@pytest.mark.gpu
def test_gpu_dynamics_limits_and_errors():
    rob17 = rtbhip.DHRobot([rtbhip.RevoluteDH(a=0.1, m=1.0) for _ in range(17)])
    from helpers import large_sizes_served
    if large_sizes_served():                            # > 16 joints: the same kernel template instantiated at run time (tests/test_large_chains_gpu.py)
        M17 = np.asarray(rob17.inertia(np.zeros(17)))
        assert M17.shape == (17, 17) and np.allclose(M17, M17.T)
    else:
        with pytest.raises(rtbhip.RtbHipError):
            rob17.inertia(np.zeros(17))                 # a box without hipRTC / the CPU replay: loud ELIMIT, no silent fallback
    assert rob17.gravload(np.zeros((3, 17))).shape == (3, 17)   # rne itself handles any n
    for n in (11, 14, 16):                                # the spilling instantiations
        rob = _long_arm(n)
        L = rob.L24()
        rng = np.random.default_rng(n)
        N = 70
        q, qd, tq = rng.uniform(-1, 1, (N, n)), rng.normal(size=(N, n)), rng.normal(size=(N, n))
        for k in (4, 11):
            if k < n:
                q[:, k] = rng.uniform(0, 0.4, N)
        k = 8
        nt.assert_allclose(rob.inertia(q)[:k], oracle.inertia_dh(L, 0, q[:k]), rtol=1e-10, atol=1e-11)
        nt.assert_allclose(rob.coriolis(q, qd)[:k], oracle.coriolis_dh(L, 0, q[:k], qd[:k]), rtol=1e-9, atol=1e-10)
        a = rob.accel(q, qd, tq)
        nt.assert_allclose(rob.rne(q, qd, a), tq, rtol=1e-7, atol=1e-7 * np.abs(tq).max())
    for n in (5, 8, 9, 10):                               # 5, 8: the non-all-revolute instantiations of the 2-wave kernels
        rob = rtbhip.DHRobot(_ten_joint_arm().links[:n])
        L = rob.L24()
        rng = np.random.default_rng(n)
        N = 130
        q, qd, tq = rng.uniform(-1, 1, (N, n)), rng.normal(size=(N, n)), rng.normal(size=(N, n))
        q[:, 4] = rng.uniform(0, 0.4, N)
        k = 20
        nt.assert_allclose(rob.inertia(q)[:k], oracle.inertia_dh(L, 0, q[:k]), rtol=1e-11, atol=1e-12)
        nt.assert_allclose(rob.coriolis(q, qd)[:k], oracle.coriolis_dh(L, 0, q[:k], qd[:k]), rtol=1e-10, atol=1e-11)
        a = rob.accel(q, qd, tq)
        nt.assert_allclose(rob.rne(q, qd, a), tq, rtol=1e-8, atol=1e-8 * np.abs(tq).max())


@pytest.mark.gpu
def test_gpu_coriolis_row_does_not_depend_on_its_tile_mates():
    """The round-2 advisor's finding (a row's result depended on its tile mates: one scheme per WAVE, chosen by the widest row).  Round 3 made
    the choice per row; round 4's two-field pass has no choice to make at all.  The property stays pinned: a
    row's C(q, qd) is bit-identical whether it is evaluated alone, among ordinary rows, or next to a row whose velocities span nine orders of
    magnitude (which used to send its whole 64-row tile to the other scheme).  DH kernel and tree kernel."""
    from rtbhip import urdf
    rng = np.random.default_rng(123)
    for rob in (rtbhip.models.DH.Panda(), urdf.load("UR5")):
        n = rob.n
        q = rng.uniform(-1.5, 1.5, (64, n))
        qd = rng.normal(size=(64, n))
        alone = np.stack([rob.coriolis(q[i], qd[i]) for i in range(64)])
        together = rob.coriolis(q, qd)
        qd_wide = qd.copy()
        qd_wide[17] = qd[17] * np.logspace(-6, 3, n)            # one wide row in the tile
        with_wide = rob.coriolis(q, qd_wide)
        keep = np.arange(64) != 17
        nt.assert_array_equal(together, alone)
        nt.assert_array_equal(with_wide[keep], alone[keep])
        nt.assert_array_equal(with_wide[17], rob.coriolis(q[17], qd_wide[17]))
        # and under a batch split (ShardedBatch row blocks): the same bits
        nt.assert_array_equal(np.concatenate([rob.coriolis(q[:23], qd_wide[:23]), rob.coriolis(q[23:], qd_wide[23:])]), with_wide)

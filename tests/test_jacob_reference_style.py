"""The reference's tests/test_jacob.py, restated against this backend (SURVEY 8c pins G6 and G7):
   jacob0 / jacobe of the ETS Puma560 at q = [.1,.2,.3,.1,.2,.3], of a 7-joint chain with a flipped Ry and SE3
   constants, and of nine six-joint Rx.Ry.Rz.tx.ty.tz chains with every flip pattern the reference tries, all
   against the numerical Jacobian of fkine (spatialmath.base.numjac with SE=3: central differences of the
   translation and vex of dR R^T).  CPU: the kernels' device code through tests/emu; GPU: through the C ABI."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import ET, ETS


def puma_ets():
    """models/ETS/Puma560.py:33-55"""
    l1, l2, l3, l4, l5, l6 = 0.672, -0.2337, 0.4318, 0.0203, 0.0837, 0.4318
    return (ET.tz(l1) * ET.Rz() * ET.ty(l2) * ET.Ry() * ET.tz(l3) * ET.tx(l4) * ET.ty(l5) * ET.Ry()
            * ET.tz(l6) * ET.Rz() * ET.Ry() * ET.Rz() * ET.tx(0.2))


def flipped0():
    """tests/test_jacob.py:101-141"""
    def se3(x, z):
        T = np.eye(4); T[0, 3] = x; T[2, 3] = z
        return ET.SE3(T)
    return ETS([ET.Rz(jindex=0), se3(-4.3624e-04, 3.6000e-01), ET.Ry(jindex=1), ET.Rz(jindex=2), se3(4.3624e-04, 4.2000e-01),
                ET.Ry(jindex=3, flip=True), ET.Rz(jindex=4), ET.tz(0.4), ET.Ry(jindex=5), ET.Rz(jindex=6), ET.tz(0.126)])


FLIPS = ["100000", "010000", "001000", "000100", "000010", "000001", "101000", "001010", "111111"]   # test_jacob.py:144-198


def flipped1(pattern):
    ctor = [ET.Rx, ET.Ry, ET.Rz, ET.tx, ET.ty, ET.tz]
    return ETS([c(flip=(b == "1")) for c, b in zip(ctor, pattern)])


def numjac(fk, q, h=1e-7):
    """spatialmath.base.numjac(f, q, SE=3): columns [dt/dq ; vex(dR/dq R^T)] by central differences."""
    q = np.asarray(q, dtype=float)
    T0 = fk(q)
    J = np.zeros((6, len(q)))
    for k in range(len(q)):
        d = np.zeros(len(q)); d[k] = h
        Tp, Tm = fk(q + d), fk(q - d)
        dT = (Tp - Tm) / (2 * h)
        J[:3, k] = dT[:3, 3]
        S = dT[:3, :3] @ T0[:3, :3].T
        J[3:, k] = [S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]]
        J[3:, k] *= 0.5
    return J


def check(ets, q, fk, j0, je):
    Jn = numjac(fk, q)
    nt.assert_array_almost_equal(j0(q), Jn)                                   # test_jacob0*, 6 decimals
    R = fk(np.asarray(q, dtype=float))[:3, :3]
    B = np.zeros((6, 6)); B[:3, :3] = R.T; B[3:, 3:] = R.T
    nt.assert_array_almost_equal(je(q), B @ Jn)                               # test_jacobe*


CASES = [("puma", puma_ets, [0.1, 0.2, 0.3, 0.1, 0.2, 0.3]), ("flipped0", flipped0, [0, -0.3, 0, -2.2, 0, 2, 0.79])]
CASES += [("flip" + p, (lambda p=p: flipped1(p)), [-0.3, 0, -2.2, 0, 2, 0.79]) for p in FLIPS]


@pytest.mark.parametrize("name,make,q", CASES, ids=[c[0] for c in CASES])
def test_emu_jacobians_equal_numerical_jacobian(name, make, q):
    import emu_harness as emu
    ets = make()
    for reg in (True, False):
        fk = lambda x: emu.kin(ets, x, want=("T",), reg=reg)[0][0]
        j0 = lambda x: emu.kin(ets, x, want=("J",), reg=reg)[1][0]
        je = lambda x: emu.kin(ets, x, want=("J",), frame=1, reg=reg)[1][0]
        check(ets, q, fk, j0, je)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,q", CASES, ids=[c[0] for c in CASES])
def test_gpu_jacobians_equal_numerical_jacobian(name, make, q):
    ets = make()
    check(ets, q, lambda x: ets.eval(x), lambda x: ets.jacob0(x), lambda x: ets.jacobe(x))
    qd = np.array([0.1, -0.2, 0.3, -0.1, 0.2, -0.3, 0.15])[:ets.n]
    # test_jacob_dot: jacob0_dot == sum_i numhess[i] qd_i
    h = 1e-6
    Jd = np.zeros((6, ets.n))
    for i in range(ets.n):
        d = np.zeros(ets.n); d[i] = h
        Jd += (ets.jacob0(np.asarray(q) + d) - ets.jacob0(np.asarray(q) - d)) / (2 * h) * qd[i]
    nt.assert_array_almost_equal(ets.jacob0_dot(q, qd), Jd, decimal=4)


# ---------------------------------------------------------------- analytical Jacobians (test_jacob.py:40-70)
REPS = ["rpy/xyz", "rpy/zyx", "eul", "exp"]
REP_CODE = {r: k for k, r in enumerate(REPS)}


def _numjac_x(ch, q, rep, tool=None):
    """numjac(lambda q: tr2x(robot.fkine(q).A, representation=rep), q) of the reference's tests."""
    from oracle import oracle
    h = 1e-6
    cols = []
    for i in range(len(q)):
        d = np.zeros(len(q)); d[i] = h
        xp = oracle.tr2x(oracle.fkine(ch, q + d, tool=tool)[0], rep)
        xm = oracle.tr2x(oracle.fkine(ch, q - d, tool=tool)[0], rep)
        dx = xp - xm
        dx[3:] = (dx[3:] + np.pi) % (2 * np.pi) - np.pi if rep != "exp" else dx[3:]
        cols.append(dx / (2 * h))
    return np.array(cols).T


@pytest.mark.parametrize("rep", REPS)
def test_oracle_and_emu_analytical_jacobian_equals_numerical(rep):
    import emu_harness as emu
    from oracle import oracle
    from helpers import chain_from_ets
    ets = puma_ets()
    ch = chain_from_ets(ets)
    q = np.array([0.1, 0.2, 0.3, 0.1, 0.2, 0.3])
    Ja = _numjac_x(ch, q, rep)
    nt.assert_array_almost_equal(oracle.jacob0_analytical(ch, q, rep)[0], Ja)                  # 6 decimals, as the reference
    nt.assert_array_almost_equal(emu.diff(ets, 3, q, axes=REP_CODE[rep])[0], Ja)
    rng = np.random.default_rng(3)
    qs = rng.uniform(-1.2, 1.2, (20, 6))
    nt.assert_allclose(emu.diff(ets, 3, qs, axes=REP_CODE[rep]), oracle.jacob0_analytical(ch, qs, rep), atol=1e-9)


@pytest.mark.gpu
def test_gpu_analytical_jacobians():
    import torch
    from oracle import oracle
    from helpers import chain_from_ets
    ets = puma_ets()
    ch = chain_from_ets(ets)
    q = np.array([0.1, 0.2, 0.3, 0.1, 0.2, 0.3])
    rng = np.random.default_rng(3)
    qs = rng.uniform(-1.2, 1.2, (300, 6))
    for rep in REPS:
        nt.assert_array_almost_equal(ets.jacob0_analytical(q, representation=rep), _numjac_x(ch, q, rep))
        Ja = ets.jacob0_analytical(qs, representation=rep)
        nt.assert_allclose(Ja, oracle.jacob0_analytical(ch, qs, rep), atol=1e-9)
        nt.assert_array_equal(ets.jacob0_analytical(torch.from_numpy(qs).cuda(), representation=rep).cpu().numpy(), Ja)
    panda = rtbhip.models.Panda()
    qp = rng.uniform(-1, 1, (10, 7))
    chp = chain_from_ets(panda.ets())
    nt.assert_allclose(panda.jacob0_analytical(qp, "eul"), oracle.jacob0_analytical(chp, qp, "eul", tool=panda.tool), atol=1e-9)
    with pytest.raises(ValueError):
        ets.jacob0_analytical(q, representation="quaternion")

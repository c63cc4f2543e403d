"""DHRobot.rne(base_wrench=True) for modified-DH chains and for robots with a base transform, checked against MOMENTUM BALANCE.

The reference routes base_wrench=True to its pure-Python rne_python (robot/DHRobot.py:1409-1412, :1458-1796).  For a standard-DH robot without a
base that formulation agrees with the compiled frne and pins the kernel (tests/test_06_reference_dh_classes.py).  Its modified-DH branch does not
agree with frne (:1640 rotates only the first term of the linear acceleration; :1711 takes the moment of F_j about p* instead of the centre of mass)
and with a base it enters gravity with the opposite sign (:1597 against :1591), so the reference holds no usable numbers for those two cases.  What
rtbhip returns there is the reference's DEFINITION -- wbase = [R_1 f_1, R_1 n_1], the wrench the base exerts on link 1 as the backward recursion of
frne holds it when it ends, turned into frame 0 (:1765-1770) -- with frne's torques.  The checker here is independent of any Newton-Euler recursion:

    force   F = sum_i m_i (c_i'' - g) + R_n f_ext
    moment  M = sum_i [ (c_i - o) x m_i (c_i'' - g) + d/dt (R_i I_i R_i^T w_i) ] + R_n n_ext + (o_n - o) x R_n f_ext

with c_i the centres of mass in frame 0 along the motion q(t) = q + qd t + qdd t^2 / 2, differentiated numerically; o is the point the first link's
moment refers to: the origin of frame 0 (standard DH), of frame 1 (modified DH).  The same checker reproduces the reference's standard-DH numbers
(first test), which pins it."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip


def _rx(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]])


def _rz(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])


def _tr(x, y, z):
    T = np.eye(4)
    T[:3, 3] = (x, y, z)
    return T


def link_A(l, mdh, qj):
    """robot/DHLink.py:1126-1189: Rz(theta) Tz(d) Tx(a) Rx(alpha), or Rx(alpha) Tx(a) Rz(theta) Tz(d) for modified DH."""
    theta, d = (l.theta, qj + l.offset) if l.sigma else (qj + l.offset, l.d)
    if mdh:
        return _rx(l.alpha) @ _tr(l.a, 0, 0) @ _rz(theta) @ _tr(0, 0, d)
    return _rz(theta) @ _tr(0, 0, d) @ _tr(l.a, 0, 0) @ _rx(l.alpha)


def frames(rob, q):
    T, out = np.eye(4), []
    for l, qj in zip(rob.links, q):
        T = T @ link_A(l, rob.mdh, qj)
        out.append(T)
    return out


def _vee(S):
    return np.array([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]]) / 2


def balance_wrench(rob, q, qd, qdd, g, fext=None, h1=2e-5, h2=2e-4):
    """(F, M) of the module docstring, frame-0 axes; g = gravitational acceleration in frame 0 (robot.gravity turned by the base)."""
    qt = lambda t: q + qd * t + 0.5 * qdd * t * t
    n = rob.n

    def state(t):          # centres of mass, angular momenta about them (frame-0 axes)
        Tm, T0, Tp = frames(rob, qt(t - h1)), frames(rob, qt(t)), frames(rob, qt(t + h1))
        c, L = [], []
        for i, l in enumerate(rob.links):
            R = T0[i][:3, :3]
            w = _vee((Tp[i][:3, :3] - Tm[i][:3, :3]) / (2 * h1) @ R.T)
            c.append(T0[i][:3, :3] @ np.asarray(l.r, dtype=float).reshape(3) + T0[i][:3, 3])
            L.append(R @ np.asarray(l.I, dtype=float).reshape(3, 3) @ R.T @ w)
        return np.array(c), np.array(L)
    cm, Lm = state(-h2)
    c0, _ = state(0.0)
    cp, Lp = state(h2)
    acc = (cp - 2 * c0 + cm) / (h2 * h2)
    Ld = (Lp - Lm) / (2 * h2)
    T = frames(rob, q)
    o = T[0][:3, 3] if rob.mdh else np.zeros(3)
    F, M = np.zeros(3), np.zeros(3)
    for i, l in enumerate(rob.links):
        Fi = l.m * (acc[i] - g)
        F += Fi
        M += np.cross(c0[i] - o, Fi) + Ld[i]
    if fext is not None:
        Rn = T[n - 1][:3, :3]
        fe, ne = Rn @ np.asarray(fext[:3], dtype=float), Rn @ np.asarray(fext[3:], dtype=float)
        F += fe
        M += ne + np.cross(T[n - 1][:3, 3] - o, fe)
    return F, M


def random_robot(rng, n, mdh, sigmas=None, base=None):
    mk = {(0, False): rtbhip.RevoluteDH, (0, True): rtbhip.RevoluteMDH, (1, False): rtbhip.PrismaticDH, (1, True): rtbhip.PrismaticMDH}
    links = []
    for k in range(n):
        s = 0 if sigmas is None else sigmas[k]
        A = rng.uniform(-0.2, 0.2, (3, 3))
        kw = dict(a=rng.uniform(-0.4, 0.4), alpha=rng.choice([0.0, np.pi / 2, -np.pi / 2, 0.3]), offset=rng.uniform(-0.3, 0.3),
                  m=rng.uniform(0.5, 4), r=rng.uniform(-0.2, 0.2, 3), I=A @ A.T + 0.05 * np.eye(3), Jm=rng.uniform(0, 1e-3), G=rng.choice([1.0, 50.0]),
                  B=rng.uniform(0, 1e-3), Tc=[rng.uniform(0, 0.2), -rng.uniform(0, 0.2)])
        if s == 0:
            kw["d"] = rng.uniform(-0.3, 0.3)
        else:
            kw["theta"] = rng.uniform(-1, 1)
            kw["qlim"] = [0.0, 1.0]
        links.append(mk[(s, mdh)](**kw))
    return rtbhip.DHRobot(links, name="rnd", base=base)


BASE = np.array([[0.0, -1, 0, 0.1], [0.6, 0, -0.8, 0], [0.8, 0, 0.6, 0.3], [0, 0, 0, 1]])


def cases():
    rng = np.random.default_rng(23)
    out = []
    for mdh in (False, True):
        # (a prismatic FIRST joint: frne itself leaves physics there -- modified DH takes the joint's rate as an angular velocity and does not turn
        #  gravity into the link frame, core/ne.c:188-206; standard DH lets gravity's x / y leak into the later links' qdd vector, core/ne.c:311 --
        #  and rtbhip returns frne's numbers, torques included: served, but outside a balance check, except standard DH under a gravity along z)
        for n, sig in ((6, None), (7, None), (5, [0, 1, 0, 0, 1]), (4, [0, 1, 1, 0] if mdh else [1, 0, 0, 1]), (11, None)):
            out.append(random_robot(rng, n, mdh, sig))
        out.append(random_robot(rng, 6, mdh, None, base=BASE))
        out.append(random_robot(rng, 5, mdh, [0, 0, 1, 0, 0], base=BASE))
    out.append(rtbhip.models.DH.Panda())
    out.append(rtbhip.models.DH.Puma560())
    return out, rng


def frame0_gravity(rob, gravity=None):
    g = np.asarray(rob.gravity if gravity is None else gravity, dtype=float).reshape(3)
    return g if rob.base is None else rob.base[:3, :3].T @ g


def check(rob, rng, call, rows=3):
    q, qd, qdd = rng.uniform(-1.5, 1.5, (3, rows, rob.n))
    for fext, gravity in ((None, None), (rng.uniform(-3, 3, 6), [0.0, 0.0, -9.0] if rob.links[0].sigma else [0.3, -0.2, -9.0])):
        tau, wb = call(rob, q, qd, qdd, gravity, fext)
        for k in range(rows):
            F, M = balance_wrench(rob, q[k], qd[k], qdd[k], frame0_gravity(rob, gravity), fext)
            scale = max(1.0, np.abs(F).max(), np.abs(M).max())
            nt.assert_allclose(wb[k, :3], F, atol=2e-5 * scale)
            nt.assert_allclose(wb[k, 3:], M, atol=2e-5 * scale)
    return tau


def test_the_balance_checker_reproduces_the_reference_for_standard_dh():
    """Pins the checker: on a standard-DH robot without a base it must return what the reference's rne_python returns as wbase."""
    from oracle import ref_classes, ref_harness
    if not (ref_classes.dh_available() and ref_harness.available()):
        pytest.skip("needs oracle/_ref")
    ns = ref_classes.load_dh(ref_harness._load("fknm"), ref_harness._load("frne"), "ref-dh-balance")
    theirs, mine = ns.Puma560(), rtbhip.models.DH.Puma560()
    rng = np.random.default_rng(2)
    q, qd, qdd = rng.uniform(-2, 2, (3, 4, 6))
    fext = rng.uniform(-3, 3, 6)
    _, wb = theirs.rne_python(q, qd, qdd, fext=fext, base_wrench=True)
    for k in range(4):
        F, M = balance_wrench(mine, q[k], qd[k], qdd[k], np.asarray(mine.gravity, dtype=float), fext)
        nt.assert_allclose(np.r_[F, M], wb[k], atol=2e-5 * max(1.0, np.abs(wb[k]).max()))


def test_kernel_body_against_momentum_balance():
    """The lane function of rtbhip_rne_base_wrench replayed on the CPU (tests/emu): standard and modified DH, prismatic joints, a base."""
    import emu_harness as emu
    robots, rng = cases()
    for rob in robots:
        def call(rob, q, qd, qdd, gravity, fext):
            return emu.rne_base_wrench(rob.L24(), rob.mdh, q, qd, qdd, rob._gravity_c(gravity), fext)
        tau = check(rob, rng, call)
        assert np.isfinite(tau).all()


@pytest.mark.gpu
def test_device_against_momentum_balance():
    robots, rng = cases()
    for rob in robots:
        def call(rob, q, qd, qdd, gravity, fext):
            tau, wb = rob.rne(q, qd, qdd, gravity=gravity, fext=fext, base_wrench=True)
            nt.assert_allclose(tau, rob.rne(q, qd, qdd, gravity=gravity, fext=fext), rtol=1e-12, atol=1e-12)        # frne's torques
            return tau, wb
        check(rob, rng, call, rows=5)
    # at rest a based robot's base carries the arm's weight along the base frame's image of "up"
    rob = robots[5]
    assert rob.base is not None and not rob.mdh
    _, w = rob.rne(np.zeros(rob.n), None, None, base_wrench=True)
    nt.assert_allclose(w[:3], -sum(l.m for l in rob.links) * (rob.base[:3, :3].T @ np.asarray(rob.gravity, dtype=float).reshape(3)), rtol=1e-12, atol=1e-12)

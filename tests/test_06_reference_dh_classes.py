"""The reference's OWN DH-side classes -- robot/DHRobot.py `DHRobot`, robot/DHLink.py `RevoluteDH` ..., robot/Dynamics.py (the mixin
behind `inertia` / `coriolis` / `gravload` / `accel`), robot/Link.py, BaseRobot.py, Robot.py and the model files models/DH/Puma560.py,
models/DH/Panda.py -- executed unmodified (oracle/ref_classes.load_dh) and bound

  * to the reference's compiled `fknm` / `frne` (oracle/_ref): the checker; pinned here on the reference's golden literals, and
  * (-m gpu) to `rtbhip.compat.fknm` / `rtbhip.compat.frne`, the plug-in shims over librtbhip.so: `DHRobot.rne` then reaches the
    MI355X through `frne.frne` exactly as it reaches the C code in a real installation (robot/DHRobot.py:1442-1451), and so do the
    Dynamics-mixin terms that are built from repeated `rne` calls.

The same loader gives the one oracle the reference has for `DHRobot.rne(base_wrench=True)`: its pure-Python `rne_python`
(robot/DHRobot.py:1458-1796), against which the kernel body (CPU replay) and the GPU are compared.
"""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from helpers import literals
from oracle import ref_classes, ref_harness

pytestmark = pytest.mark.skipif(not (ref_classes.dh_available() and ref_harness.available()),
                                reason="needs oracle/_ref (the reference's compiled extensions and byte-compiled classes)")

QN = np.array([0, np.pi / 4, np.pi, 0, np.pi / 4, 0])          # models/DH/Puma560.py qn


def ref_dh():
    return ref_classes.load_dh(ref_harness._load("fknm"), ref_harness._load("frne"), "ref-dh")


def test_reference_dhrobot_on_reference_extensions_reproduces_the_goldens():
    ns = ref_dh()
    puma = ns.Puma560()
    assert type(puma).__mro__[1].__module__ == "roboticstoolbox.robot.DHRobot" and puma.n == 6 and not puma.mdh
    z, o = np.zeros(6), np.ones(6)
    L = literals()
    # tests/test_DHRobot.py:1036-1062 (G9)
    for (qd, qdd), key in (((z, z), "tr0"), ((z, o), "tr1"), ((o, o), "tr2"), ((o, z), "tr3")):
        nt.assert_allclose(puma.rne(QN, qd, qdd), L["G9_puma_rne_" + key], atol=5e-4)
    nt.assert_allclose(puma.rne(QN, o, o, gravity=[0, 0, 0]), L["G9_puma_rne_tr4"], atol=5e-4)
    nt.assert_allclose(puma.rne(QN, z, z, fext=[1, 2, 3, 1, 2, 3]), L["G9_puma_rne_tr5"], atol=5e-4)
    # the Dynamics-mixin literals of tests/test_DHRobot.py (inertia, coriolis, gravload, itorque, accel at qn)
    nt.assert_allclose(puma.inertia(QN), L["D_puma_inertia"], atol=5e-4)
    nt.assert_allclose(puma.coriolis(QN, np.array(L["D_puma_coriolis_qd"])), L["D_puma_coriolis"], atol=5e-4)
    nt.assert_allclose(puma.gravload(QN), L["D_puma_gravload"], atol=5e-4)
    nt.assert_allclose(puma.itorque(QN, np.array(L["D_puma_itorque_qdd"])), L["D_puma_itorque"], atol=5e-4)
    nt.assert_allclose(puma.accel(QN, np.array(L["D_puma_accel_qd"]), np.array(L["D_puma_accel_torque"])), L["D_puma_accel"], atol=5e-3)
    # the two formulations of the reference agree for a standard-DH arm ...
    rng = np.random.default_rng(3)
    q, qd, qdd = rng.uniform(-2, 2, (3, 4, 6))
    nt.assert_allclose(puma.rne_python(q, qd, qdd), puma.rne(q, qd, qdd), rtol=1e-10, atol=1e-10)
    # ... and not for a modified-DH one (robot/DHRobot.py:1649): the reason rtbhip.DHRobot.rne(base_wrench=True) refuses MDH chains
    panda = ns.PandaDH()
    q7, qd7, qdd7 = rng.uniform(-1, 1, (3, 7))
    assert np.abs(panda.rne_python(q7, qd7, qdd7) - panda.rne(q7, qd7, qdd7)).max() > 1.0


def test_model_files_of_the_reference_equal_the_mirrored_models():
    """models/DH/Puma560.py and models/DH/Panda.py, executed: their links carry the parameters rtbhip.models.DH has."""
    ns = ref_dh()
    for theirs, mine in ((ns.Puma560(), rtbhip.models.DH.Puma560()), (ns.PandaDH(), rtbhip.models.DH.Panda())):
        assert theirs.n == mine.n and bool(theirs.mdh) == bool(mine.mdh)
        for a, b in zip(theirs.links, mine.links):
            nt.assert_allclose([a.d, a.a, a.alpha, a.offset, a.m, a.Jm, a.G, a.B], [b.d, b.a, b.alpha, b.offset, b.m, b.Jm, b.G, b.B], atol=0)
            nt.assert_allclose(a.r, b.r, atol=0); nt.assert_allclose(a.I, b.I, atol=0); nt.assert_allclose(a.Tc, b.Tc, atol=0)
        nt.assert_allclose(theirs.gravity, mine.gravity, atol=0)
        tool = mine.tool if mine.tool is not None else np.eye(4)
        nt.assert_allclose(theirs.tool.A, tool, atol=1e-15)
        nt.assert_allclose(theirs.qlim, mine.qlim, atol=1e-15)


def random_dh_pair(ns, rng, sigmas):
    """The same random standard-DH robot as the reference's DHRobot and as rtbhip.DHRobot: six joints (the only size the reference's
    base-wrench call works for), full inertia tensors, off-origin centres of mass, gears, viscous and Coulomb friction."""
    theirs, mine = [], []
    for s in sigmas:
        A = rng.uniform(-0.2, 0.2, (3, 3))
        kw = dict(a=rng.uniform(-0.5, 0.5), alpha=rng.choice([0.0, np.pi / 2, -np.pi / 2, 0.3]), offset=rng.uniform(-0.3, 0.3),
                  m=rng.uniform(0.5, 5), r=rng.uniform(-0.3, 0.3, 3), I=A @ A.T + 0.05 * np.eye(3), Jm=rng.uniform(0, 1e-3),
                  G=rng.choice([-60.0, 1.0, 80.0]), B=rng.uniform(0, 1e-3), Tc=[rng.uniform(0, 0.3), -rng.uniform(0, 0.3)])
        if s == 0:
            kw["d"] = rng.uniform(-0.4, 0.4)
            theirs.append(ns.RevoluteDH(**kw)); mine.append(rtbhip.RevoluteDH(**kw))
        else:
            kw["theta"] = rng.uniform(-1, 1)
            kw["qlim"] = [0.0, 1.0]
            kw["offset"] = 0.0          # rne_python takes a prismatic joint's extension as q alone (robot/DHRobot.py:1612), frne as q + offset
            theirs.append(ns.PrismaticDH(**kw)); mine.append(rtbhip.PrismaticDH(**kw))
    return ns.DHRobot(theirs, name="rnd"), rtbhip.DHRobot(mine, name="rnd")


def base_wrench_cases(ns):
    rng = np.random.default_rng(17)
    out = [(ns.Puma560(), rtbhip.models.DH.Puma560())]
    for sig in ([0] * 6, [0, 1, 0, 0, 1, 0], [1, 0, 0, 1, 0, 0]):
        out.append(random_dh_pair(ns, rng, sig))
    return out, rng


def test_base_wrench_kernel_body_against_the_references_rne_python():
    """rne_python(base_wrench=True) of the reference (robot/DHRobot.py:1765-1770) == the kernel body replayed on the CPU."""
    import emu_harness as emu
    ns = ref_dh()
    cases, rng = base_wrench_cases(ns)
    for theirs, mine in cases:
        q, qd, qdd = rng.uniform(-2, 2, (3, 9, 6))
        for fext in (None, rng.uniform(-3, 3, 6)):
            tau_r, wb_r = theirs.rne_python(q, qd, qdd, fext=fext, base_wrench=True)
            nt.assert_allclose(theirs.rne(q, qd, qdd, fext=fext), tau_r, rtol=1e-9, atol=1e-9)          # its two formulations agree
            tau_e, wb_e = emu.rne_base_wrench(mine.L24(), False, q, qd, qdd, mine._gravity_c(None), fext)
            nt.assert_allclose(tau_e, tau_r, rtol=1e-9, atol=1e-9)
            nt.assert_allclose(wb_e, wb_r, rtol=1e-9, atol=1e-9)
    # the single-row form: flat (n,) and (6,)
    theirs, mine = cases[0]
    t1, w1 = theirs.rne(QN, np.ones(6), np.ones(6), base_wrench=True)
    assert t1.shape == (6,) and w1.shape == (6,)


def test_byte_compiled_dh_classes_are_what_the_gpu_box_loads(monkeypatch):
    """oracle/_ref/pyref/*.pyc (make -f oracle/Makefile refpy) of the DH side load without /root/reference and behave as the .py files."""
    import os
    if not all(os.path.exists(ref_classes._pyc(f)) for f in ref_classes.FILES + ref_classes.DH_FILES):
        pytest.skip("oracle/_ref/pyref not built")
    monkeypatch.setattr(ref_classes, "REF_PKG", "/nonexistent")
    ns = ref_classes.load_dh(ref_harness._load("fknm"), ref_harness._load("frne"), "ref-dh-pyc")
    ref = ref_dh()
    assert ns.DHRobot is not ref.DHRobot and ns.DHRobot.__module__ == "roboticstoolbox.robot.DHRobot"
    q, qd, qdd = np.random.default_rng(0).uniform(-2, 2, (3, 6))
    nt.assert_array_equal(ns.Puma560().rne(q, qd, qdd), ref.Puma560().rne(q, qd, qdd))
    nt.assert_array_equal(ns.Puma560().fkine(q).A, ref.Puma560().fkine(q).A)


# ------------------------------------------------------------------------------------------------ on the device
@pytest.mark.gpu
def test_base_wrench_on_the_device():
    ns = ref_dh()
    cases, rng = base_wrench_cases(ns)
    for theirs, mine in cases:
        q, qd, qdd = rng.uniform(-2, 2, (3, 33, 6))
        for fext in (None, rng.uniform(-3, 3, 6)):
            tau_r, wb_r = theirs.rne_python(q, qd, qdd, fext=fext, base_wrench=True)
            tau, wb = mine.rne(q, qd, qdd, fext=fext, base_wrench=True)
            assert tau.shape == (33, 6) and wb.shape == (33, 6)
            nt.assert_allclose(tau, tau_r, rtol=1e-9, atol=1e-9)
            nt.assert_allclose(wb, wb_r, rtol=1e-9, atol=1e-9)
            nt.assert_allclose(tau, mine.rne(q, qd, qdd, fext=fext), rtol=1e-12, atol=1e-12)           # the fast kernel's torques
    theirs, mine = cases[0]
    t1, w1 = mine.rne(QN, np.ones(6), np.ones(6), base_wrench=True)
    tr, wr = theirs.rne(QN, np.ones(6), np.ones(6), base_wrench=True)
    assert t1.shape == (6,) and w1.shape == (6,)
    nt.assert_allclose(t1, tr, rtol=1e-10); nt.assert_allclose(w1, wr, rtol=1e-10)
    # device tensors in -> device tensors out
    import torch
    qt, qdt, qddt = (torch.from_numpy(x).cuda() for x in rng.uniform(-2, 2, (3, 1000, 6)))
    tau_d, wb_d = mine.rne(qt, qdt, qddt, base_wrench=True)
    assert tau_d.is_cuda and wb_d.shape == (1000, 6)
    tau_h, wb_h = mine.rne(qt.cpu().numpy(), qdt.cpu().numpy(), qddt.cpu().numpy(), base_wrench=True)
    nt.assert_array_equal(wb_d.cpu().numpy(), wb_h)
    # modified DH and robots with a base: the reference's rne_python disagrees with its own frne there (robot/DHRobot.py:1640, :1711, :1597); served by
    # the definition with frne's torques, against momentum balance in tests/test_base_wrench_balance.py
    panda = rtbhip.models.DH.Panda()
    q7p = rng.uniform(-1, 1, (4, 7))
    tau_p, wb_p = panda.rne(q7p, q7p, q7p, base_wrench=True)
    nt.assert_allclose(tau_p, panda.rne(q7p, q7p, q7p), rtol=1e-12, atol=1e-12)
    assert wb_p.shape == (4, 6) and np.isfinite(wb_p).all()
    based = rtbhip.DHRobot(mine.links, base=np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0.2], [0, 0, 0, 1.0]]))
    tau_b, wb_b = based.rne(QN, QN, QN, base_wrench=True)
    nt.assert_allclose(tau_b, based.rne(QN, QN, QN), rtol=1e-12, atol=1e-12)
    # seven joints: served with a (N, 6) wrench (the reference's own call raises there: wbase is allocated (N, n))
    links7 = [rtbhip.RevoluteDH(d=0.1 * k, a=0.05 * k, alpha=(-1) ** k * np.pi / 2, m=1.0 + k, r=[0.01 * k, 0.02, 0.03]) for k in range(7)]
    arm7 = rtbhip.DHRobot(links7)
    q7 = rng.uniform(-1, 1, (5, 7))
    tau7, wb7 = arm7.rne(q7, q7, q7, base_wrench=True)
    nt.assert_allclose(tau7, arm7.rne(q7, q7, q7), rtol=1e-12, atol=1e-12)
    assert wb7.shape == (5, 6) and np.isfinite(wb7).all()
    # at rest the base carries the arm's weight: f_z = g * sum(m) in frame 0
    _, wrest = arm7.rne(q7, None, None, base_wrench=True)
    nt.assert_allclose(wrest[:, 2], 9.81 * sum(l.m for l in links7), rtol=1e-12)


@pytest.mark.gpu
def test_reference_dhrobot_on_the_gpu_shims():
    """The reference's DHRobot / Dynamics-mixin methods with `roboticstoolbox.frne` = rtbhip.compat.frne and
    `roboticstoolbox.fknm` = rtbhip.compat.fknm == the same classes on the reference's compiled extensions."""
    import rtbhip.compat
    gpu = ref_classes.load_dh(rtbhip.compat.fknm, rtbhip.compat.frne, "rtbhip-dh")
    ref = ref_dh()
    assert gpu.DHRobot is not ref.DHRobot and gpu.frne is rtbhip.compat.frne
    rng = np.random.default_rng(5)
    for make in ("Puma560", "PandaDH"):
        g, r = getattr(gpu, make)(), getattr(ref, make)()
        n = g.n
        q, qd, qdd = rng.uniform(-1.5, 1.5, (3, 12, n))
        nt.assert_allclose(g.rne(q, qd, qdd), r.rne(q, qd, qdd), rtol=1e-11, atol=1e-11)               # trajectory form, DHRobot.py:1442-1451
        nt.assert_allclose(g.rne(q[0], qd[0], qdd[0], fext=[1, 2, 3, 1, 2, 3]), r.rne(q[0], qd[0], qdd[0], fext=[1, 2, 3, 1, 2, 3]),
                           rtol=1e-11, atol=1e-11)
        nt.assert_allclose(g.rne(q[1], qd[1], qdd[1], gravity=[0, 0, 0]), r.rne(q[1], qd[1], qdd[1], gravity=[0, 0, 0]), rtol=1e-11, atol=1e-11)
        # Dynamics mixin: every term is a loop of rne calls (robot/Dynamics.py:704-922), each of them now a kernel launch
        nt.assert_allclose(g.gravload(q[2]), r.gravload(q[2]), rtol=1e-11, atol=1e-11)
        nt.assert_allclose(g.inertia(q[3]), r.inertia(q[3]), rtol=1e-10, atol=1e-11)
        nt.assert_allclose(g.coriolis(q[4], qd[4]), r.coriolis(q[4], qd[4]), rtol=1e-9, atol=1e-10)
        nt.assert_allclose(g.itorque(q[5], qdd[5]), r.itorque(q[5], qdd[5]), rtol=1e-10, atol=1e-11)
        nt.assert_allclose(g.accel(q[6], qd[6], qdd[6]), r.accel(q[6], qd[6], qdd[6]), rtol=1e-8, atol=1e-9)
        # the kinematics of the DH classes: DHRobot.fkine is the reference's Python product of link matrices (no extension involved);
        # its ets() lowering runs through fknm
        nt.assert_allclose(g.fkine(q[7]).A, r.fkine(q[7]).A, atol=1e-14)
        nt.assert_allclose(g.ets().eval(q), r.ets().eval(q), atol=1e-12)
        nt.assert_allclose(g.ets().eval(q[7]), r.fkine(q[7]).A, atol=1e-12)
        nt.assert_allclose(g.ets().jacob0(q[8]), r.jacob0(q[8]), atol=1e-11)
        # and the batched mirror of the same robot agrees with the reference's class row by row
        mine = rtbhip.models.DH.Puma560() if make == "Puma560" else rtbhip.models.DH.Panda()
        nt.assert_allclose(mine.rne(q, qd, qdd), r.rne(q, qd, qdd), rtol=1e-11, atol=1e-11)
        nt.assert_allclose(mine.inertia(q)[3], r.inertia(q[3]), rtol=1e-10, atol=1e-11)
        nt.assert_allclose(mine.coriolis(q, qd)[4], r.coriolis(q[4], qd[4]), rtol=1e-9, atol=1e-10)
        nt.assert_allclose(mine.fkine(q)[7], r.fkine(q[7]).A, atol=1e-12)
        nt.assert_allclose(mine.jacob0(q)[8], r.jacob0(q[8]), atol=1e-11)
        nt.assert_allclose(mine.jacobe(q)[8], r.jacobe(q[8]), atol=1e-11)

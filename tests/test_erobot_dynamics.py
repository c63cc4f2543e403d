"""Dynamics-mixin terms of ETS robots (inertia, coriolis, gravload, itorque, accel): reference robot/Dynamics.py:424-509, 704-922,
1407-1465, which `Robot` inherits and builds from repeated `Robot.rne` calls (SURVEY 8f-2 on the robots of 8f-1).

Oracle: oracle/erobot.py restates the mixin's loops over erobot_rne (itself pinned on the reference's closed forms and cross-pinned on
its compiled frne, tests/test_erobot_rne.py).  The restated loops are pinned on the reference's OWN `DynamicsMixin` methods --
robot/Dynamics.py executed unmodified (oracle/ref_classes.load_dh) on a stand-in robot whose `rne` is erobot_rne -- bit for bit.
The kernel body (tests/emu: tree_device.h's tree_dyn_lane) and the GPU are compared with that oracle."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import ERobot, urdf
from oracle import erobot as oer, ref_classes, ref_harness
from test_erobot_rne import random_tree, dfs, spong


def serial_case(seed, n_links):
    """(product ERobot, oracle link list in the reference's order) of a random robot with at most nine joints."""
    rng = np.random.default_rng(seed)
    while True:
        prod, orc = random_tree(rng, n_links=n_links)
        rob = ERobot(prod)
        if 1 <= rob.n <= 9:
            return rob, dfs(orc), rng


def test_restated_mixin_loops_equal_the_references_own_mixin():
    if not (ref_classes.dh_available() and ref_harness.available()):
        pytest.skip("needs oracle/_ref")
    ns = ref_classes.load_dh(ref_harness._load("fknm"), ref_harness._load("frne"), "ref-dh")
    Mixin = ns.mods["Dynamics"].DynamicsMixin
    rob, links, rng = serial_case(3, 6)
    n = rob.n

    class OnOracle(Mixin):
        """what the mixin needs of a robot: n, gravity, rne, nofriction (robot/Dynamics.py:744-763, 811-861, 483-505, 905-922, 1451-1465)"""
        gravity = np.array([0.3, -0.2, -9.81])

        def __init__(self): self.n = n
        def rne(self, q, qd, qdd, gravity=None):
            out = oer.erobot_rne(links, q, qd, qdd, self.gravity if gravity is None else gravity)
            return out[0] if out.shape[0] == 1 else out
        def nofriction(self, coulomb=True, viscous=False): return self

    duck = OnOracle()
    q, qd, tq = rng.uniform(-2, 2, (3, n)), rng.normal(size=(3, n)), rng.normal(size=(3, n))
    nt.assert_array_equal(duck.inertia(q), oer.erobot_inertia(links, q))
    nt.assert_array_equal(duck.coriolis(q, qd), oer.erobot_coriolis(links, q, qd))
    nt.assert_array_equal(duck.accel(q, qd, tq), oer.erobot_accel(links, q, qd, tq, duck.gravity))
    nt.assert_array_equal(duck.gravload(q), oer.erobot_rne(links, q, 0 * q, 0 * q, duck.gravity))
    nt.assert_array_equal(duck.itorque(q, tq), oer.erobot_rne(links, q, 0 * q, tq, (0, 0, 0)))
    assert duck.inertia(q[0]).shape == (n, n) and duck.accel(q[0], qd[0], tq[0]).shape == (n,)       # one row: the flat forms


@pytest.mark.parametrize("seed,n_links", [(0, 3), (1, 5), (2, 8), (4, 10), (5, 7)])
def test_emu_kernel_body_vs_oracle(seed, n_links):
    import emu_harness as emu
    rob, links, rng = serial_case(seed, n_links)
    n = rob.n
    q, qd, tq = rng.uniform(-2, 2, (5, n)), rng.normal(size=(5, n)), rng.normal(size=(5, n))
    g = np.array([0.5, -0.3, -9.81])
    recs = rob.group_table()
    M = oer.erobot_inertia(links, q)
    scale = max(1.0, np.abs(M).max())
    nt.assert_allclose(emu.tree_dyn(recs, 0, q), M, rtol=0, atol=1e-12 * scale)
    Cw = oer.erobot_coriolis(links, q, qd)
    nt.assert_allclose(emu.tree_dyn(recs, 1, q, qd), Cw, rtol=0, atol=1e-12 * max(1.0, np.abs(Cw).max()))
    if np.linalg.cond(M).max() < 1e8:          # (a random tree may carry a massless tip: M singular, nothing to compare)
        want = oer.erobot_accel(links, q, qd, tq, g)
        nt.assert_allclose(emu.tree_dyn(recs, 2, q, qd, tq, g), want, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(want).max()))


def test_emu_coriolis_scale_cases_and_rest():
    """velocities from 1e-9 to 1e9, rows whose velocities span 2^40 (the reference's own scheme, per row), a row at rest (exact zeros)"""
    import emu_harness as emu
    rob, links, rng = serial_case(1, 5)
    n = rob.n
    base = rng.normal(size=n)
    qd = np.stack([base * 1e-9, base * 1e9, base, np.r_[base[:-1], base[-1] * 2.0 ** 40], np.zeros(n)])
    q = rng.uniform(-2, 2, (len(qd), n))
    got, want = emu.tree_dyn(rob.group_table(), 1, q, qd), oer.erobot_coriolis(links, q, qd)
    for i in range(len(qd)):
        nt.assert_allclose(got[i], want[i], rtol=0, atol=1e-12 * np.abs(want[i]).max())
    assert not got[-1].any()


def renumbered_case(seed, n_links):
    """serial_case with the joints numbered by hand in a shuffled order: q / qd / qdd / torque are read by jindex, torques come back in group
    order (Robot.py:1830-1893) -- so the reference's `inertia` is the symmetric group-ordered matrix with its ROWS permuted, and `accel` solves
    with that matrix.  (Every URDF robot and every automatically numbered robot is in group order; this is the hand-numbered rest.)"""
    rng = np.random.default_rng(seed)
    while True:
        prod, orc = random_tree(rng, n_links=n_links)
        n = sum(1 for l in prod if l.isjoint)
        if 3 <= n <= 9:
            break
    perm = rng.permutation(n)
    while np.array_equal(perm, np.arange(n)):
        perm = rng.permutation(n)
    by_name, k = {}, 0
    for l in dfs(orc):
        if any(not isinstance(it, np.ndarray) and (len(it) == 1 or it[1] is None) for it in l["ets"]):
            by_name[l["name"]] = int(perm[k]); k += 1
    for l in orc:
        if l["name"] in by_name:
            l["jindex"] = by_name[l["name"]]
    prod2 = []
    for l in prod:
        parent = None if l.parent is None else [p for p in prod2 if p.name == l.parent.name][0]
        prod2.append(rtbhip.Link(l.ets, jindex=by_name.get(l.name), m=l.m, r=l.r, parent=parent, name=l.name))
    return ERobot(prod2), orc, rng              # (hand-numbered links keep the order they were given in: BaseRobot.py:353-370 `orlinks = links`)


@pytest.mark.parametrize("seed,n_links", [(11, 6), (12, 8), (13, 5)])
def test_emu_hand_numbered_joints(seed, n_links):
    import emu_harness as emu
    rob, links, rng = renumbered_case(seed, n_links)
    n = rob.n
    recs = rob.group_table()
    assert [r["jindex"] for r in recs] != list(range(n))
    q, qd, tq = rng.uniform(-2, 2, (4, n)), rng.normal(size=(4, n)), rng.normal(size=(4, n))
    g = np.array([0.5, -0.3, -9.81])
    nt.assert_allclose(emu.tree_rne(recs, q, qd, tq, g), oer.erobot_rne(links, q, qd, tq, g), rtol=1e-10, atol=1e-10)
    M = oer.erobot_inertia(links, q)
    nt.assert_allclose(emu.tree_dyn(recs, 0, q), M, rtol=0, atol=1e-12 * max(1.0, np.abs(M).max()))
    Cw = oer.erobot_coriolis(links, q, qd)
    nt.assert_allclose(emu.tree_dyn(recs, 1, q, qd), Cw, rtol=0, atol=1e-12 * max(1.0, np.abs(Cw).max()))
    if np.linalg.cond(M).max() < 1e8:
        want = oer.erobot_accel(links, q, qd, tq, g)
        nt.assert_allclose(emu.tree_dyn(recs, 2, q, qd, tq, g), want, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_links", [(11, 6), (12, 8), (13, 5)])
def test_gpu_hand_numbered_joints(seed, n_links):
    rob, links, rng = renumbered_case(seed, n_links)
    n = rob.n
    q, qd, tq = rng.uniform(-2, 2, (70, n)), rng.normal(size=(70, n)), rng.normal(size=(70, n))
    g = np.array([0.5, -0.3, -9.81])
    rob.gravity = g
    k = slice(0, 4)
    M = oer.erobot_inertia(links, q[k])
    nt.assert_allclose(rob.inertia(q)[k], M, rtol=0, atol=1e-12 * max(1.0, np.abs(M).max()))
    Cw = oer.erobot_coriolis(links, q[k], qd[k])
    nt.assert_allclose(rob.coriolis(q, qd)[k], Cw, rtol=0, atol=1e-12 * max(1.0, np.abs(Cw).max()))
    nt.assert_allclose(rob.rne(q, qd, tq)[k], oer.erobot_rne(links, q[k], qd[k], tq[k], g), rtol=1e-10, atol=1e-10)
    if np.linalg.cond(M).max() < 1e8:
        want = oer.erobot_accel(links, q[k], qd[k], tq[k], g)
        nt.assert_allclose(rob.accel(q, qd, tq)[k], want, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(want).max()))


def big_case(seed, n):
    """a random branched robot with exactly n joints (10..20: the largest instantiations of the tree dynamics kernels)"""
    rng = np.random.default_rng(seed)
    while True:
        prod, orc = random_tree(rng, n_links=n + int(rng.integers(0, 5)))
        rob = ERobot(prod)
        if rob.n == n:
            return rob, dfs(orc), rng


@pytest.mark.parametrize("n", [10, 11, 12, 13, 14, 16, 18, 20])
def test_emu_ten_to_twenty_joints(n):
    import emu_harness as emu
    rob, links, rng = big_case(40 + n, n)
    recs = rob.group_table()
    q, qd, tq = rng.uniform(-2, 2, (2, n)), rng.normal(size=(2, n)), rng.normal(size=(2, n))
    g = np.array([0.5, -0.3, -9.81])
    M = oer.erobot_inertia(links, q)
    nt.assert_allclose(emu.tree_dyn(recs, 0, q), M, rtol=0, atol=1e-12 * max(1.0, np.abs(M).max()))
    Cw = oer.erobot_coriolis(links, q, qd)
    nt.assert_allclose(emu.tree_dyn(recs, 1, q, qd), Cw, rtol=0, atol=1e-12 * max(1.0, np.abs(Cw).max()))
    if np.linalg.cond(M).max() < 1e8:
        want = oer.erobot_accel(links, q, qd, tq, g)
        nt.assert_allclose(emu.tree_dyn(recs, 2, q, qd, tq, g), want, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [10, 11, 12, 13, 15, 16, 17, 19, 20])
def test_gpu_ten_to_twenty_joints(n):
    rob, links, rng = big_case(40 + n, n)
    q, qd, tq = rng.uniform(-2, 2, (70, n)), rng.normal(size=(70, n)), rng.normal(size=(70, n))
    g = np.array([0.5, -0.3, -9.81])
    rob.gravity = g
    k = slice(0, 2)
    M, Cm = rob.inertia(q), rob.coriolis(q, qd)
    Mo = oer.erobot_inertia(links, q[k])
    nt.assert_allclose(M[k], Mo, rtol=0, atol=1e-12 * max(1.0, np.abs(Mo).max()))
    Co = oer.erobot_coriolis(links, q[k], qd[k])
    nt.assert_allclose(Cm[k], Co, rtol=0, atol=1e-12 * max(1.0, np.abs(Co).max()))
    tau = rob.rne(q, qd, tq)                                  # all 70 rows: rne(q, qd, qdd) = M qdd + C qd + g
    nt.assert_allclose(np.einsum("nij,nj->ni", M, tq) + np.einsum("nij,nj->ni", Cm, qd) + rob.gravload(q), tau,
                       rtol=0, atol=1e-10 * max(1.0, np.abs(tau).max()))
    if np.linalg.cond(M).max() < 1e8:
        qdd = rob.accel(q, qd, tq)
        nt.assert_allclose(rob.rne(q, qd, qdd), tq, rtol=0, atol=1e-8 * max(1.0, np.abs(tq).max()))


def urdf_pairs():
    out = []
    for name in ("UR5", "Panda"):
        rob = urdf.load(name)
        out.append((name, rob))
    return out


# ------------------------------------------------------------------------------------------------ on the device
@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_links", [(0, 3), (1, 5), (2, 8), (4, 10), (5, 7), (6, 9)])
def test_gpu_random_trees_vs_oracle(seed, n_links):
    rob, links, rng = serial_case(seed, n_links)
    n = rob.n
    q, qd, tq = rng.uniform(-2, 2, (70, n)), rng.normal(size=(70, n)), rng.normal(size=(70, n))
    g = np.array([0.5, -0.3, -9.81])
    rob.gravity = g
    k = slice(0, 6)
    M = rob.inertia(q)
    Mo = oer.erobot_inertia(links, q[k])
    nt.assert_allclose(M[k], Mo, rtol=0, atol=1e-12 * max(1.0, np.abs(Mo).max()))
    nt.assert_allclose(M, np.swapaxes(M, 1, 2), rtol=0, atol=1e-11 * max(1.0, np.abs(M).max()))          # symmetric to rounding
    Cm, Co = rob.coriolis(q, qd), oer.erobot_coriolis(links, q[k], qd[k])
    nt.assert_allclose(Cm[k], Co, rtol=0, atol=1e-12 * max(1.0, np.abs(Co).max()))
    nt.assert_allclose(rob.gravload(q)[k], oer.erobot_rne(links, q[k], 0 * q[k], 0 * q[k], g), rtol=1e-10, atol=1e-10)
    nt.assert_allclose(rob.itorque(q, tq)[k], oer.erobot_rne(links, q[k], 0 * q[k], tq[k], (0, 0, 0)), rtol=1e-10, atol=1e-10)
    # consistency over the whole batch (all 70 rows, two tiles): rne(q, qd, qdd) = M qdd + C qd + g
    tau = rob.rne(q, qd, tq)
    nt.assert_allclose(np.einsum("nij,nj->ni", M, tq) + np.einsum("nij,nj->ni", Cm, qd) + rob.gravload(q), tau,
                       rtol=0, atol=1e-10 * max(1.0, np.abs(tau).max()))
    if np.linalg.cond(M).max() < 1e8:
        qdd = rob.accel(q, qd, tq)
        want = oer.erobot_accel(links, q[k], qd[k], tq[k], g)
        nt.assert_allclose(qdd[k], want, rtol=1e-8, atol=1e-8 * max(1.0, np.abs(want).max()))
        nt.assert_allclose(rob.rne(q, qd, qdd), tq, rtol=0, atol=1e-8 * max(1.0, np.abs(tq).max()))       # forward then inverse dynamics
    # shapes of the one-row forms (robot/Dynamics.py:760-763) and device tensors
    assert rob.inertia(q[0]).shape == (n, n) and rob.coriolis(q[0], qd[0]).shape == (n, n) and rob.accel(q[0], qd[0], tq[0]).shape == (n,)
    import torch
    qt, qdt = torch.from_numpy(q).cuda(), torch.from_numpy(qd).cuda()
    Mt = rob.inertia(qt)
    assert Mt.is_cuda
    nt.assert_array_equal(Mt.cpu().numpy(), M)
    nt.assert_array_equal(rob.coriolis(qt, qdt).cpu().numpy(), Cm)


@pytest.mark.gpu
def test_gpu_urdf_arms_and_the_dh_cross_pin():
    """URDF UR5 / Panda: the terms are consistent with the robot's own rne; and for a DH arm whose links are point masses the tree
    kernel's terms equal the DH kernel's (two formulations of the reference, robot/Robot.py:1797 drops the inertia tensor)."""
    rng = np.random.default_rng(8)
    for name in ("UR5", "Panda"):
        rob = urdf.load(name)
        n = rob.n
        if n > 12:
            continue
        q, qd, qdd = rng.uniform(-1.5, 1.5, (200, n)), rng.normal(size=(200, n)), rng.normal(size=(200, n))
        M, Cm, G = rob.inertia(q), rob.coriolis(q, qd), rob.gravload(q)
        tau = rob.rne(q, qd, qdd)
        nt.assert_allclose(np.einsum("nij,nj->ni", M, qdd) + np.einsum("nij,nj->ni", Cm, qd) + G, tau, rtol=0, atol=1e-10 * np.abs(tau).max())
        if np.linalg.cond(M).max() < 1e10:
            nt.assert_allclose(rob.rne(q, qd, rob.accel(q, qd, tau)), tau, rtol=0, atol=1e-7 * np.abs(tau).max())
    from test_erobot_rne import dh_point_mass_arm
    from oracle import chains
    for robot in ("puma560", "panda_dh"):
        tab = getattr(chains, robot)()
        tab.tool = None
        L, orc, prod = dh_point_mass_arm(tab, rng)
        er = ERobot(prod)
        dh = rtbhip.DHRobot([rtbhip.dh.DHLink(alpha=r[0], a=r[1], theta=r[2], d=r[3], sigma=int(r[4]), mdh=bool(tab.mdh), offset=r[5], m=r[6],
                                              r=r[7:10], G=r[20]) for r in L])
        n = dh.n
        q, qd, tq = rng.uniform(-1.5, 1.5, (40, n)), rng.normal(size=(40, n)), rng.normal(size=(40, n))
        nt.assert_allclose(er.rne(q, qd, tq), dh.rne(q, qd, tq), rtol=0, atol=1e-10 * np.abs(dh.rne(q, qd, tq)).max())
        nt.assert_allclose(er.inertia(q), dh.inertia(q), rtol=0, atol=1e-10 * np.abs(dh.inertia(q)).max())
        nt.assert_allclose(er.coriolis(q, qd), dh.coriolis(q, qd), rtol=0, atol=1e-10 * np.abs(dh.coriolis(q, qd)).max())
        nt.assert_allclose(er.accel(q, qd, tq), dh.accel(q, qd, tq), rtol=1e-7, atol=1e-7 * np.abs(dh.accel(q, qd, tq)).max())


@pytest.mark.gpu
def test_gpu_yumi_whole_and_without_grippers_and_the_limit_beyond_thirty_two_joints():
    """YuMi -- two 7-joint arms off one body, 14 joints without its grippers, 18 with them: the widest robot of the fleet -- through the 14- and
    18-joint instantiations: the terms are consistent with the robot's own rne on every row.  Trees of 21 .. 32 joints get their kernels at run
    time (tests/test_large_chains_gpu.py); a robot of more than RTBHIP_MAX_JOINTS = 32 joints is refused loudly (no fallback)."""
    rob = urdf.load("YuMi")
    arms = ("gripper_r_base", "gripper_l_base")
    assert rob.erobot(arms).n == 14 and rob.n == 18
    rng = np.random.default_rng(14)
    for exclude, n in ((arms, 14), ((), 18)):
        q, qd, tq = rng.uniform(-1.5, 1.5, (70, n)), rng.normal(size=(70, n)), rng.normal(size=(70, n))
        M, Cm, tau = rob.inertia(q, exclude=exclude), rob.coriolis(q, qd, exclude=exclude), rob.rne(q, qd, tq, exclude=exclude)
        nt.assert_allclose(M, np.swapaxes(M, 1, 2), rtol=0, atol=1e-11 * np.abs(M).max())
        nt.assert_allclose(np.einsum("nij,nj->ni", M, tq) + np.einsum("nij,nj->ni", Cm, qd) + rob.gravload(q, exclude=exclude), tau,
                           rtol=0, atol=1e-10 * np.abs(tau).max())
        if np.linalg.cond(M).max() < 1e8:
            nt.assert_allclose(rob.rne(q, qd, rob.accel(q, qd, tq, exclude=exclude), exclude=exclude), tq, rtol=0, atol=1e-7 * np.abs(tq).max())
    big, _, _ = big_case(3, 33)
    with pytest.raises(rtbhip.RtbHipError):
        big.inertia(np.zeros(big.n))

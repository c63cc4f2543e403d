"""ETS-robot inverse dynamics (reference Robot.rne robot/Robot.py:1704-1903).
Pins: the closed-form two-link cases of the reference's tests/test_ERobot.py:101-274 (plain, with a
static middle link, with a static link carrying the mass).  Everything else is differential against the
oracle restatement (oracle/erobot.py), which SURVEY 8c marks "parity unpinned" beyond those cases."""
from math import pi, sin, cos

import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import ET, ETS, Link, ERobot, urdf
from oracle import erobot as oer, chains

Z2 = np.zeros(2)


def spong(variant):
    """(product ERobot, oracle link list) of the three robots of tests/test_ERobot.py:101-274"""
    if variant == 0:
        l1 = Link(ets=ETS(ET.Ry()), m=1, r=[0.5, 0, 0], name="l1")
        l2 = Link(ets=ETS(ET.tx(1)) * ET.Ry(), m=1, r=[0.5, 0, 0], parent=l1, name="l2")
        prod = [l1, l2]
        orc = [dict(name="l1", parent=None, ets=[("Ry",)], m=1, r=[0.5, 0, 0]),
               dict(name="l2", parent="l1", ets=[("tx", 1.0), ("Ry",)], m=1, r=[0.5, 0, 0])]
    elif variant == 1:
        l1 = Link(ets=ETS(ET.Ry()), m=1, r=[0.5, 0, 0], name="l1")
        l2 = Link(ets=ETS(), m=0, r=[0, 0, 0], parent=l1, name="l2")
        l3 = Link(ets=ETS(ET.tx(1)) * ET.Ry(), m=1, r=[0.5, 0, 0], parent=l2, name="l3")
        prod = [l1, l2, l3]
        orc = [dict(name="l1", parent=None, ets=[("Ry",)], m=1, r=[0.5, 0, 0]),
               dict(name="l2", parent="l1", ets=[], m=0, r=[0, 0, 0]),
               dict(name="l3", parent="l2", ets=[("tx", 1.0), ("Ry",)], m=1, r=[0.5, 0, 0])]
    else:
        l1 = Link(ets=ETS(ET.Ry()), m=1, r=[0.5, 0, 0], name="l1")
        l2 = Link(ets=ETS(ET.tx(1)), m=1, r=[0.5, 0, 0], parent=l1, name="l2")
        l3 = Link(ets=ETS(ET.Ry()), m=0, r=[0, 0, 0], parent=l2, name="l3")
        prod = [l1, l2, l3]
        orc = [dict(name="l1", parent=None, ets=[("Ry",)], m=1, r=[0.5, 0, 0]),
               dict(name="l2", parent="l1", ets=[("tx", 1.0)], m=1, r=[0.5, 0, 0]),
               dict(name="l3", parent="l2", ets=[("Ry",)], m=0, r=[0, 0, 0])]
    return ERobot(prod, name="simple"), orc


def closed_form_cases():
    """(q, qd, qdd, gravity, expected) exactly as the reference asserts them."""
    g = [0, 0, -9.81]
    out = [(np.zeros(2), Z2, Z2, g, np.r_[-2, -0.5] * 9.81),
           (np.array([0.0, -pi / 2.0]), Z2, Z2, g, np.r_[-1.5, 0] * 9.81),
           (np.array([-pi / 2, pi / 2]), Z2, Z2, g, np.r_[-0.5, -0.5] * 9.81),
           (np.array([-pi / 2, 0]), Z2, Z2, g, np.r_[0, 0] * 9.81)]
    q = np.array([0, -pi / 2])
    h = -0.5 * sin(q[1])
    z = [0, 0, 0]
    out += [(q, np.array([0.0, 0]), Z2, z, np.r_[0, 0] * h), (q, np.array([1.0, 0]), Z2, z, np.r_[0, -1] * h),
            (q, np.array([0.0, 1]), Z2, z, np.r_[1, 0] * h), (q, np.array([1.0, 1]), Z2, z, np.r_[3, -1] * h)]
    d11, d12, d22 = 1.5 + cos(q[1]), 0.25 + 0.5 * cos(q[1]), 0.25
    out += [(q, Z2, np.array([0.0, 0]), z, np.r_[0, 0]), (q, Z2, np.array([1.0, 0]), z, np.r_[d11, d12]),
            (q, Z2, np.array([0.0, 1]), z, np.r_[d12, d22]), (q, Z2, np.array([1.0, 1]), z, np.r_[d11 + d12, d12 + d22])]
    return out


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_oracle_pinned_on_reference_closed_forms(variant):
    _, orc = spong(variant)
    for q, qd, qdd, g, want in closed_form_cases():
        nt.assert_array_almost_equal(oer.erobot_rne(orc, q, qd, qdd, g)[0], want)


def random_tree(rng, n_links=9):
    """A branched robot exercising every joint kind, flips, static links (with mass) and SE3 constants."""
    axes = ["Rx", "Ry", "Rz", "tx", "ty", "tz"]
    prod, orc = [], []
    for i in range(n_links):
        parent = None if i == 0 else int(rng.integers(max(0, i - 3), i))
        items, ets = [], ETS()
        T = chains.elementary("Rz", rng.uniform(-1, 1)) @ chains.elementary("tx", rng.uniform(-.3, .3)) @ chains.elementary("Rx", rng.uniform(-1, 1))
        items.append(T); ets = ets * ET.SE3(T)
        if rng.uniform() < 0.5:
            a, v = axes[int(rng.integers(0, 6))], float(rng.uniform(-0.5, 0.5))
            items.append((a, v)); ets = ets * getattr(ET, a)(v)
        if rng.uniform() < 0.75 or i == n_links - 1:
            a, fl = axes[int(rng.integers(0, 6))], bool(rng.uniform() < 0.3)
            items.append((a, None, fl)); ets = ets * getattr(ET, a)(flip=fl)
        m, r = float(rng.uniform(0, 3)), rng.uniform(-0.3, 0.3, 3)
        name = "k%d" % i
        prod.append(Link(ets=ets, m=m, r=r, parent=(prod[parent] if parent is not None else None), name=name))
        orc.append(dict(name=name, parent=(None if parent is None else "k%d" % parent), ets=items, m=m, r=r))
    return prod, orc


def dfs(orc):
    """links in the depth-first order the reference's _sort_links produces"""
    kids = {l["name"]: [] for l in orc}
    for l in orc:
        if l["parent"] is not None:
            kids[l["parent"]].append(l)
    out, stack = [], [orc[0]]
    while stack:
        l = stack.pop()
        out.append(l)
        stack.extend(reversed(kids[l["name"]]))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_emu_kernel_body_and_group_table_vs_oracle(seed):
    """tests/emu runs tree_device.h + tree.cpp on the CPU; the product's group table (erobot.py) feeds it."""
    import emu_harness as emu
    rng = np.random.default_rng(seed)
    prod, orc = random_tree(rng, n_links=int(rng.integers(2, 11)))
    rob = ERobot(prod)
    if rob.n == 0:
        pytest.skip("no joints drawn")
    q, qd, qdd = rng.uniform(-2, 2, (7, rob.n)), rng.normal(size=(7, rob.n)), rng.normal(size=(7, rob.n))
    g = rng.normal(size=3) * 5
    got = emu.tree_rne(rob.group_table(), q, qd, qdd, g)
    want = oer.erobot_rne(dfs(orc), q, qd, qdd, g)
    nt.assert_allclose(got, want, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(want).max()))


def test_emu_closed_forms_all_variants():
    import emu_harness as emu
    for variant in (0, 1, 2):
        rob, _ = spong(variant)
        for q, qd, qdd, g, want in closed_form_cases():
            nt.assert_array_almost_equal(emu.tree_rne(rob.group_table(), q, qd, qdd, g)[0], want)


def test_link_and_robot_construction_rules():
    with pytest.raises(ValueError):
        Link(ets=ETS(ET.Ry()) * ET.tx(1))                     # joint must be last
    a, b = Link(ets=ETS(ET.Rz()), name="a"), Link(ets=ETS(ET.Rz()), name="a")
    with pytest.raises(ValueError):
        ERobot([a, b])                                        # duplicate names
    l1, l2, l3 = Link(ets=ETS(ET.Rz())), Link(ets=ETS(ET.tx(1)) * ET.Ry()), Link(ets=ETS(ET.tz(0.2)))
    r = ERobot([l1, l2, l3])                                  # no parents given: a chain in list order
    assert l2.parent is l1 and l3.parent is l2 and r.n == 2 and [l.jindex for l in r.links] == [0, 1, None]
    assert r.link_groups() == [[0], [1]]                      # the trailing static link joins no group (Robot.py:1781-1789)
    assert r.ets().n == 2 and r.ets().m == 4


@pytest.mark.gpu
def test_gpu_closed_forms_and_shapes():
    for variant in (0, 1, 2):
        rob, _ = spong(variant)
        for q, qd, qdd, g, want in closed_form_cases():
            rob.gravity = np.array(g, dtype=float)
            tau = rob.rne(q, qd, qdd)
            assert tau.shape == (2,)
            nt.assert_array_almost_equal(tau, want)
        rob.gravity = np.array([0, 0, -9.81])
        tau2 = rob.rne(np.zeros((3, 2)), np.zeros((3, 2)), np.zeros((3, 2)))
        assert tau2.shape == (3, 2)
        nt.assert_array_almost_equal(tau2[2] / 9.81, np.r_[-2, -0.5])


@pytest.mark.gpu
@pytest.mark.parametrize("seed,N", [(0, 1), (1, 65), (2, 1000), (3, 4097), (4, 64), (5, 300)])
def test_gpu_random_trees_vs_oracle(seed, N):
    import torch
    rng = np.random.default_rng(100 + seed)
    prod, orc = random_tree(rng, n_links=int(rng.integers(3, 13)))
    rob = ERobot(prod)
    if rob.n == 0:
        pytest.skip("no joints drawn")
    q, qd, qdd = rng.uniform(-2, 2, (N, rob.n)), rng.normal(size=(N, rob.n)), rng.normal(size=(N, rob.n))
    g = rng.normal(size=3) * 5
    tau = rob.rne(q, qd, qdd, gravity=g).reshape(N, rob.n)
    k = min(N, 100)
    want = oer.erobot_rne(dfs(orc), q[:k], qd[:k], qdd[:k], g)
    nt.assert_allclose(tau[:k], want, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(want).max()))
    # linear in qdd at fixed (q, qd): tau(qdd1 + qdd2) - tau(0) == (tau(qdd1) - tau(0)) + (tau(qdd2) - tau(0))
    z = np.zeros_like(q)
    t0, t1, t2 = (rob.rne(q, qd, x, gravity=g).reshape(N, rob.n) for x in (z, qdd, 2 * qdd))
    nt.assert_allclose(t2 - t0, 2 * (t1 - t0), rtol=1e-9, atol=1e-9 * max(1.0, np.abs(t1).max()))
    qt, qdt, qddt = (torch.from_numpy(x).cuda() for x in (q, qd, qdd))
    nt.assert_array_equal(rob.rne(qt, qdt, qddt, gravity=g).cpu().numpy().reshape(N, rob.n), tau)


URDF_CASES = [("UR5", ()), ("KinovaGen3", ()), ("UR10", ()), ("AL5D", ()), ("Fetch", ()), ("px100", ()), ("Mico", ()), ("YuMi", ()),
              ("YuMi", ("gripper_r_base", "gripper_l_base"))]


def _urdf_case(name, exclude):
    r = urdf.load(name)
    exclude = tuple(e for e in exclude if e in r.linkdict)
    er = r.erobot(exclude)
    orc = []
    for l in er.links:                                        # er.links is already in depth-first order
        items = [(e.axis, None, e.isflip) if e.isjoint else np.array(e.T) for e in l.ets]
        orc.append(dict(name=l.name, parent=None if l.parent is None else l.parent.name, ets=items, m=l.m, r=l.r))
    rng = np.random.default_rng(9)
    N = 200
    q, qd, qdd = rng.uniform(-2, 2, (N, er.n)), rng.normal(size=(N, er.n)), rng.normal(size=(N, er.n))
    return r, er, orc, exclude, q, qd, qdd


@pytest.mark.parametrize("name,exclude", URDF_CASES)
def test_emu_urdf_robots_vs_oracle(name, exclude):
    """URDF robots with <inertial> data: erobot() link tree -> group table -> kernel body (tests/emu) vs oracle."""
    import emu_harness as emu
    r, er, orc, exclude, q, qd, qdd = _urdf_case(name, exclude)
    if er.n > 24:
        pytest.skip("more than 24 link groups")
    assert sum(l.m for l in er.links) > 0
    got = emu.tree_rne(er.group_table(), q[:10], qd[:10], qdd[:10], [0, 0, -9.81])
    want = oer.erobot_rne(orc, q[:10], qd[:10], qdd[:10])
    nt.assert_allclose(got, want, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(want).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name,exclude", URDF_CASES)
def test_gpu_urdf_robots_vs_oracle(name, exclude):
    r, er, orc, exclude, q, qd, qdd = _urdf_case(name, exclude)
    if er.n > 24:
        with pytest.raises(rtbhip.RtbHipError):
            r.rne(q, qd, qdd, exclude=exclude)                 # loud ELIMIT, no fallback
        return
    tau = r.rne(q, qd, qdd, exclude=exclude)
    want = oer.erobot_rne(orc, q[:40], qd[:40], qdd[:40])
    nt.assert_allclose(tau[:40], want, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(want).max()))


# ------------------------------------------------------------------------------------------------ cross-pin on the compiled frne
# The reference has TWO inverse dynamics: the compiled recursive Newton-Euler of DH robots (core/ne.c:62-493 behind frne.frne)
# and the spatial-vector Robot.rne of ETS robots (robot/Robot.py:1704-1903), which keeps only SpatialInertia(m, r) of every
# link -- no inertia tensor (:1797), no motor inertia, no friction.  For a DH arm whose links are POINT MASSES (I = 0,
# Jm = B = Tc = 0) the two must therefore give the same torques.  frne is runnable here (oracle/_ref), so this pins
# oracle/erobot.py -- and through it k_tree_rne -- on the reference's own binary for real 6- and 7-joint arms, beyond the two-link
# closed forms of tests/test_ERobot.py.
def dh_point_mass_arm(tab, rng):
    """(L24 with point masses, oracle link list, product links) of a DH table lowered the way Robot(DHRobot.ets()) cuts it:
    one link per joint, constants in front.  The mass of DH link j sits at r_j in DH frame {j}; the ETS link frame j is the frame
    right after joint j's motion, so r' = C_j r_j with C_j the constants that follow the joint inside the DH link (standard
    DH: tz(d) tx(a) Rx(alpha); modified DH: none, the joint closes the link)."""
    L = tab.L24().copy()
    L[:, 10:19] = 0.0                    # inertia tensor
    L[:, 19] = 0.0                       # Jm
    L[:, 21:24] = 0.0                    # B, Tc+, Tc-
    L[:, 7:10] = rng.uniform(-0.3, 0.3, (tab.n, 3))
    L[:, 6] = rng.uniform(0.2, 5.0, tab.n)
    ch = tab.ets()
    items, j = [], -1
    segs, after = [], []                 # per joint: items of its link; constants following it before the next joint
    for i in range(ch.m):
        if int(ch.kind[i]) == 6:
            T = ch.consts[i].reshape(4, 4)
            items.append(T)
            if j >= 0:
                after[j] = after[j] @ T
        else:
            items.append((("Rx", "Ry", "Rz", "tx", "ty", "tz")[int(ch.kind[i])], None, bool(ch.flip[i])))
            segs.append(items)
            items = []
            after.append(np.eye(4))
            j += 1
    orc, prod = [], []
    for j, seg in enumerate(segs):
        C = np.eye(4) if tab.mdh else after[j]
        r = C[:3, :3] @ L[j, 7:10] + C[:3, 3]
        orc.append(dict(name="l%d" % j, parent=None if j == 0 else "l%d" % (j - 1), ets=seg, m=L[j, 6], r=r))
        ets = ETS()
        for it in seg:
            ets = ets * (ET.SE3(it) if isinstance(it, np.ndarray) else getattr(ET, it[0])(flip=it[2]))
        prod.append(Link(ets=ets, m=L[j, 6], r=r, parent=prod[-1] if prod else None, name="l%d" % j))
    return L, orc, prod


@pytest.mark.parametrize("robot", ["puma560", "panda_dh"])
def test_oracle_erobot_rne_equals_compiled_frne_for_point_mass_dh_arms(robot):
    from oracle import ref_harness
    import emu_harness as emu
    if not ref_harness.available():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    tab = getattr(chains, robot)()
    tab.tool = None
    L, orc, prod = dh_point_mass_arm(tab, rng)
    N = 40
    q, qd, qdd = rng.uniform(-2.5, 2.5, (N, tab.n)), rng.normal(size=(N, tab.n)), rng.normal(size=(N, tab.n))
    for g in ([0, 0, -9.81], [1.0, -2.0, 3.0], [0, 0, 0]):
        ref = ref_harness.RefRNE(L, tab.mdh, gravity=g)
        want = ref.rne(q, qd, qdd)                                        # core/ne.c through frne.frne
        ref.delete()
        scale = max(1.0, np.abs(want).max())
        got = oer.erobot_rne(orc, q, qd, qdd, g)                          # robot/Robot.py:1704-1903 restated
        assert np.abs(got - want).max() <= 1e-12 * scale, np.abs(got - want).max()
        ker = emu.tree_rne(ERobot(prod).group_table(), q, qd, qdd, g)     # the kernel body, replayed on the CPU
        assert np.abs(ker - want).max() <= 1e-9 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["puma560", "panda_dh"])
def test_gpu_tree_rne_equals_compiled_frne_for_point_mass_dh_arms(robot):
    from oracle import ref_harness
    rng = np.random.default_rng(5)
    tab = getattr(chains, robot)()
    tab.tool = None
    L, orc, prod = dh_point_mass_arm(tab, rng)
    rob = ERobot(prod)
    N = 300
    q, qd, qdd = rng.uniform(-2.5, 2.5, (N, tab.n)), rng.normal(size=(N, tab.n)), rng.normal(size=(N, tab.n))
    tau = rob.rne(q, qd, qdd)
    if ref_harness.available():
        ref = ref_harness.RefRNE(L, tab.mdh, gravity=[0, 0, -9.81])
        want = ref.rne(q, qd, qdd)
        ref.delete()
    else:
        want = oer.erobot_rne(orc, q, qd, qdd)
    assert np.abs(tau - want).max() <= 1e-9 * max(1.0, np.abs(want).max())

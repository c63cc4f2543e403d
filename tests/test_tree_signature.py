"""Structure signatures of dynamics trees (csrc/tree_device.h: kTreeSigUR; tree.cpp: tree_signature; the dispatch of tree_kernels.hip / tree_dyn_kernels.hip).

The group constants of a URDF robot are mostly not general rotations; for the signature this build has instantiations for -- the UR family's --
`k_tree_rne` / `k_tree_dyn` multiply by every constant in the form of its class, drop the cross-product terms of the zero translation
components and compile the tree bookkeeping (parents, branch slots, prismatic joints) away.  The forms are the general recursion's with exact
zeros dropped: the specialised kernels must agree with the general ones (rtbhip_tune "tree_sig" = 0) to rounding, and with the oracle
(oracle/erobot.py, pinned on the reference's own mixin: test_erobot_dynamics.py) exactly as the general ones do.

`-m "not gpu"`: the kernel bodies on the CPU replay (tests/emu mirrors the launcher's dispatch); `-m gpu`: the kernels themselves."""
import ctypes as C

import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import urdf
from oracle import erobot as oer

UR = ("UR3", "UR5", "UR10")


IBX8 = ("px150", "rx150", "rx200", "vx300", "wx200", "wx250")


def _signature(rob, which="ur"):
    return _signature_of_table(rob.erobot(()).group_table(), which)


def _signature_of_table(recs, which="ur"):
    import emu_harness as emu
    from rtbhip._lib import rtbhip_tree_group
    arr = (rtbhip_tree_group * len(recs))()
    for k, r in enumerate(recs):
        arr[k].parent, arr[k].kind, arr[k].flip, arr[k].jindex = r["parent"], r["kind"], r["flip"], r["jindex"]
        arr[k].T[:] = list(np.ascontiguousarray(r["T"]).reshape(16))
        arr[k].m = r["m"]
        arr[k].h[:] = list(r["h"])
        arr[k].I[:] = list(r["I"])
    f = emu.lib().emu_tree_signature
    f.argtypes, f.restype = [C.POINTER(rtbhip_tree_group), C.c_int32], C.c_uint64
    g = getattr(emu.lib(), "emu_tree_signature_" + which)
    g.restype = C.c_uint64
    return f(arr, len(recs)), g()


def _fields(sig, n):
    return [((sig >> (7 * j)) & 15, (sig >> (7 * j + 4)) & 7) for j in range(n)]


@pytest.mark.parametrize("name", UR)
def test_ur_family_has_the_instantiated_signature(name):
    sig, want = _signature(urdf.load(name))
    assert sig == want, (_fields(sig, 6), _fields(want, 6), hex(sig >> 56))
    assert sig >> 63 == 1 and (sig >> 56) & 1 == 1                    # present, plain serial chain of revolute joints


@pytest.mark.parametrize("name", IBX8)
def test_interbotix_arms_of_eight_groups_share_one_signature(name):
    sig, want = _signature(urdf.load(name), "ibx8")
    assert sig == want, (_fields(sig, 8), _fields(want, 8))
    assert (sig >> 56) & 1 == 0                                        # prismatic fingers: not a plain chain
    # ... and one bookkeeping word (tree_device.h: TreeTopo): six revolute groups in series, then the two prismatic fingers
    import emu_harness as emu
    from rtbhip._lib import rtbhip_tree_group
    recs = urdf.load(name).erobot(()).group_table()
    arr = (rtbhip_tree_group * len(recs))()
    for k, r in enumerate(recs):
        arr[k].parent, arr[k].kind, arr[k].flip, arr[k].jindex = r["parent"], r["kind"], r["flip"], r["jindex"]
        arr[k].T[:] = list(np.ascontiguousarray(r["T"]).reshape(16))
        arr[k].m = r["m"]
        arr[k].h[:] = list(r["h"])
        arr[k].I[:] = list(r["I"])
    f, g = emu.lib().emu_tree_topology, emu.lib().emu_tree_topology_ibx8
    f.argtypes = [C.POINTER(rtbhip_tree_group), C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    g.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    hi, lo, whi, wlo = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    assert f(arr, len(recs), C.byref(hi), C.byref(lo)) == 0
    g(C.byref(whi), C.byref(wlo))
    assert (hi.value, lo.value) == (whi.value, wlo.value)
    t = (hi.value << 64) | lo.value
    assert [((t >> (12 * j)) & 15) - 1 for j in range(8)] == [r["parent"] for r in recs] == list(range(-1, 7))
    assert [(t >> (12 * j + 4)) & 1 for j in range(8)] == [0] * 6 + [1, 1]


def test_longer_trees_carry_a_second_class_word_and_take_their_instantiation():
    """9 .. 16 groups: the fields of groups 8 .. 15 sit in a second word; up to 10 groups the bookkeeping word exists too.  The replay library dispatches
    exactly as the launcher (tests/emu/emu_misc.cpp): with the switch on these robots run the instantiated bodies (the equality test below covers them)."""
    import emu_harness as emu
    from rtbhip._lib import rtbhip_tree_group
    f2 = emu.lib().emu_tree_signature2
    f2.argtypes, f2.restype = [C.POINTER(rtbhip_tree_group), C.c_int32], C.c_uint64
    want = {"vx300s": (0x80970504b58de641, 0x8000000000000001), "wx250s": (0x80970504b58de641, 0x8000000000000001),
            "Fetch": (0x80592d65ca380581, 0x800000000000164b), "Mico": (0x8063c18f0c09f047, 0x80000000000018f0)}
    for name, (w1, w2) in want.items():
        recs = urdf.load(name).erobot(()).group_table()
        arr = (rtbhip_tree_group * len(recs))()
        for k, r in enumerate(recs):
            arr[k].parent, arr[k].kind, arr[k].flip, arr[k].jindex = r["parent"], r["kind"], r["flip"], r["jindex"]
            arr[k].T[:] = list(np.ascontiguousarray(r["T"]).reshape(16))
            arr[k].m = r["m"]
            arr[k].h[:] = list(r["h"])
            arr[k].I[:] = list(r["I"])
        sig, _ = _signature_of_table(recs)
        assert (sig, f2(arr, len(recs))) == (w1, w2), name
        assert (sig >> 56) & 1 == 0


def test_print_signatures_script_reports_the_words_the_kernels_dispatch_on():
    """scripts/print_signatures.py is the maintainer's side of `a robot = its constants + one dispatch line`: its output for an instantiated robot must
    carry exactly the words tree_device.h holds."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "print_signatures.py"), "UR5", "wx250", "Mico"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stdout
    assert "0x81000264b3145041 plain" in text and "0x80032e0a042de641" in text and "hi 0x8000000001701600 lo 0x5004003002001000" in text
    assert "second word 0x80000000000018f0" in text and "RneSig 0xe00047a99a2c7ea9" in text


def test_other_robots_do_not():
    for name in ("Panda", "Puma560"):
        sig, want = _signature(urdf.load(name))
        assert sig != want
    sig, _ = _signature(urdf.load("YuMi"))                            # more groups than a signature describes
    assert sig == 0


def _terms(rob, q, qd, tq, g):
    return {"rne": np.asarray(rob.rne(q, qd, tq, gravity=g)), "gravload": np.asarray(rob.gravload(q, gravity=g)),
            "itorque": np.asarray(rob.itorque(q, tq)), "inertia": np.asarray(rob.inertia(q)), "coriolis": np.asarray(rob.coriolis(q, qd)),
            "accel": np.asarray(rob.accel(q, qd, tq, gravity=g))}


def _both(rob, q, qd, tq, g):
    out = {}
    try:
        for s in (1, 0):
            rtbhip.tune("tree_sig", s)
            out[s] = _terms(rob, q, qd, tq, g)
    finally:
        rtbhip.tune("tree_sig", 1)
    return out


def _check(name, N, seed):
    rob = urdf.load(name)
    rng = np.random.default_rng(seed)
    n = rob.n
    q, qd, tq = rng.uniform(-3, 3, (N, n)), rng.normal(size=(N, n)), rng.normal(size=(N, n))
    g = np.array([0.4, -0.3, -9.81])
    out = _both(rob, q, qd, tq, g)
    regular = np.linalg.cond(out[0]["inertia"]) < 1e8       # (the UR3's point-mass inertia matrix is singular: the reference's quirk, Robot.py:1793-1800)
    for k in out[1]:
        a, b = out[1][k], out[0][k]
        if k == "accel":
            a, b = a[regular], b[regular]
        assert np.isfinite(b).all() and b.size or k == "accel"
        tol = (1e-9 if k == "accel" else 1e-12) * max(1.0, np.abs(b).max() if b.size else 1.0)
        nt.assert_allclose(a, b, rtol=0, atol=tol, err_msg=k)
    return rob, q, qd, tq, g, out[1]


LONGER = ("vx300s", "wx250s", "Fetch", "Mico")          # 9 .. 10 groups: a second class word; the Mico with a real branch point (one slot)


@pytest.mark.parametrize("name", UR + ("wx250", "px150", "px100") + LONGER)
def test_signature_kernels_equal_the_general_kernels_on_the_cpu_replay(name):
    import cpu_backend
    with cpu_backend.installed():
        rob, q, qd, tq, g, got = _check(name, 40, 11)
    er = rob.erobot(())
    links = [dict(name=l.name, parent=None if l.parent is None else l.parent.name, m=l.m, r=l.r,
                  ets=[(e.axis, None, e.isflip) if e.isjoint else np.array(e.T) for e in l.ets]) for l in er.links]
    k = slice(0, 6)
    M = oer.erobot_inertia(links, q[k])
    terms = [("rne", oer.erobot_rne(links, q[k], qd[k], tq[k], g)), ("inertia", M), ("coriolis", oer.erobot_coriolis(links, q[k], qd[k]))]
    if np.linalg.cond(M).max() < 1e8:
        terms.append(("accel", oer.erobot_accel(links, q[k], qd[k], tq[k], g)))
    for term, want in terms:
        nt.assert_allclose(got[term][k], want, rtol=0, atol=(1e-8 if term == "accel" else 1e-11) * max(1.0, np.abs(want).max()), err_msg=term)


def test_switch_leaves_a_branched_robot_with_another_signature_alone():
    import cpu_backend
    with cpu_backend.installed():
        rob = _branched_arm()                                          # five groups, a real branch point: no instantiation
        sig, _ = _signature_of_table(rob.group_table())
        assert sig >> 63 == 1 and (sig >> 56) & 1 == 0
        rng = np.random.default_rng(3)
        q, qd, tq = rng.uniform(-3, 3, (10, rob.n)), rng.normal(size=(10, rob.n)), rng.normal(size=(10, rob.n))
        out = _both(rob, q, qd, tq, np.array([0, 0, -9.81]))
    for k in out[1]:
        nt.assert_array_equal(out[1][k], out[0][k])


def _branched_arm():
    from rtbhip import ET, ETS, Link, ERobot
    rng = np.random.default_rng(77)
    base = Link(ETS([ET.SE3(_random_se3(rng)), ET.Rz()]), name="b", m=1.0, r=[0.1, 0, 0])
    a1 = Link(ETS([ET.SE3(_random_se3(rng)), ET.Ry()]), name="a1", parent=base, m=0.5, r=[0, 0.1, 0])
    a2 = Link(ETS([ET.SE3(_random_se3(rng)), ET.Rx()]), name="a2", parent=a1, m=0.4, r=[0, 0, 0.1])
    c1 = Link(ETS([ET.SE3(_random_se3(rng)), ET.Rz()]), name="c1", parent=base, m=0.7, r=[0.05, 0, 0.02])
    c2 = Link(ETS([ET.SE3(_random_se3(rng)), ET.tz()]), name="c2", parent=c1, m=0.3, r=[0, 0.03, 0])
    return ERobot([base, a1, a2, c1, c2])


def _random_serial_arm(n, seed):
    """a serial chain of n revolute joints about random axes with random constants and point masses (rtbhip.ERobot of ETS links)"""
    from rtbhip import ET, ETS, Link, ERobot
    rng = np.random.default_rng(seed)
    links, parent = [], None
    for j in range(n):
        const = ET.SE3(_random_se3(rng))
        joint = [ET.Rx, ET.Ry, ET.Rz][int(rng.integers(3))](flip=bool(rng.integers(2)))
        l = Link(ETS([const, joint]), name="l%d" % j, parent=parent, m=float(rng.uniform(0.5, 3)), r=rng.normal(size=3) * 0.1)
        links.append(l)
        parent = l
    return ERobot(links)


def _random_se3(rng):
    from scipy.spatial.transform import Rotation
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(rng.normal(size=3)).as_matrix()
    T[:3, 3] = rng.normal(size=3) * 0.3
    return T


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 8, 9])
def test_plain_chain_instantiation_serves_any_serial_revolute_arm(n):
    """kTreeSigPlainChain (tree_device.h): a serial chain of up to eight revolute joints numbered in order takes the straight-line kernels whatever its
    constants are -- same arithmetic as the general kernels, the bookkeeping compiled away: results agree to rounding; nine joints: general."""
    import cpu_backend
    rob = _random_serial_arm(n, 40 + n)
    sig, _ = _signature_of_table(rob.group_table())
    assert ((sig >> 56) & 1 == 1) == (n <= 8)
    rng = np.random.default_rng(n)
    q, qd, tq = rng.uniform(-3, 3, (12, n)), rng.normal(size=(12, n)), rng.normal(size=(12, n))
    with cpu_backend.installed():
        out = _both(rob, q, qd, tq, np.array([0.2, 0.1, -9.81]))
    for k in out[1]:
        b = out[0][k]
        nt.assert_allclose(out[1][k], b, rtol=0, atol=(1e-9 if k == "accel" else 1e-12) * max(1.0, np.abs(b).max()), err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", UR + ("wx250", "rx150", "px100") + LONGER)
def test_gpu_signature_kernels_equal_the_general_kernels(name):
    _check(name, 5000, 12)

"""-m "not gpu": the C-ABI library loads, exports every symbol include/rtbhip.h declares, its host
logic (chain compiler, argument checking, shard arithmetic) behaves, and compute entry points fail
LOUDLY -- not silently fall back -- when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rtbhip.h")).read()
    declared = set(re.findall(r"\b(rtbhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 19
    handle = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), "librtbhip.so does not export " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert handle.rtbhip_version() >= 100


def _info(ets):
    n, m, qw = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(_lib.lib().rtbhip_chain_info(ets._handle(), C.byref(n), C.byref(m), C.byref(qw)))
    return n.value, m.value, qw.value


def test_chain_compiler_counts_and_folding():
    p = rtbhip.models.Panda().ets()
    assert (p.n, p.m) == (7, 22)
    assert _info(p) == (7, 22, 7)          # sparse elementary constants are kept specialised
    # identity constants vanish; two general constants fold into one
    A = np.eye(4); A[:3, :3] = [[0, -1, 0], [0, 0, -1], [1, 0, 0]]; A[:3, 3] = [1, 2, 3]
    e = rtbhip.ET.SE3(A) * rtbhip.ET.SE3(A) * rtbhip.ET.tx(0.0) * rtbhip.ET.Rz() * rtbhip.ET.Rx(0.0)
    assert _info(e) == (1, 5, 1)
    # jindex: all-or-none, explicit indices define the q width
    e = rtbhip.ET.Rz(jindex=2) * rtbhip.ET.tx(1.0) * rtbhip.ET.Ry(jindex=0)
    assert _info(e) == (2, 3, 3)
    with pytest.raises(ValueError):
        (rtbhip.ET.Rz(jindex=1) * rtbhip.ET.Rz()).jindices


def test_q_width_of_a_branch_can_be_the_robots():
    """A branch of a tree robot is evaluated on the ROBOT's q (reference Robot.jacob0(q, start, end), robot/Robot.py:1974-1981):
    rtbhip_chain_set_q_width widens the row pitch beyond max(jindex)+1; narrower than the chain needs, or beyond 256, is refused."""
    e = rtbhip.ET.Rz(jindex=2) * rtbhip.ET.tx(1.0) * rtbhip.ET.Ry(jindex=0)
    assert e.q_width == 3
    e.q_width = 9
    assert _info(e) == (2, 3, 9) and e.q_width == 9
    lib = _lib.lib()
    assert lib.rtbhip_chain_set_q_width(e._handle(), 2) == -1 and b"outside [3, 256]" in lib.rtbhip_last_error()
    assert lib.rtbhip_chain_set_q_width(e._handle(), 257) == -1
    assert lib.rtbhip_chain_set_q_width(12345, 4) == -1
    with pytest.raises(ValueError):
        e.q_width = 2
    e.q_width = None
    assert _info(e) == (2, 3, 3)
    from rtbhip import urdf
    y = urdf.load("YuMi")
    r, l = y.ets(end="gripper_r_finger_r", compact=False), y.ets(end="gripper_l_finger_l", compact=False)
    assert r.q_width == l.q_width == y.n == 18 and _info(r)[2] == 18
    # depth first from the body, as the reference numbers (BaseRobot._sort_links): the right arm 0..6 and its two fingers 7, 8, the left arm 9..15 and
    # its fingers 16, 17 -- a finger each here
    assert sorted(r.jindices) == list(range(8)) and sorted(l.jindices) == list(range(9, 16)) + [17]


def test_chain_create_rejects_bad_input():
    lib = _lib.lib()
    h = C.c_uint64(0)
    arr = (_lib.rtbhip_et * 1)()
    arr[0].kind = 9
    assert lib.rtbhip_chain_create(arr, 1, None, C.byref(h)) == -1
    assert b"unknown transform kind" in lib.rtbhip_last_error()
    arr[0].kind = _lib.ET_CONST          # non-affine constant
    for k in range(16):
        arr[0].T[k] = 1.0
    assert lib.rtbhip_chain_create(arr, 1, None, C.byref(h)) == -1
    assert lib.rtbhip_chain_create(None, 3, None, C.byref(h)) == -1
    assert lib.rtbhip_chain_destroy(123456789) == -1
    assert lib.rtbhip_dyn_create(None, 3, 0, C.byref(h)) == -1
    L = np.zeros((2, 24)); L[0, 4] = 5.0
    assert lib.rtbhip_dyn_create(L.ctypes.data_as(C.c_void_p), 2, 0, C.byref(h)) == -1
    assert lib.rtbhip_dyn_create(L.ctypes.data_as(C.c_void_p), 2, 3, C.byref(h)) == -1


def test_default_and_explicit_qlim():
    e = rtbhip.ET.Rz() * rtbhip.ET.tx() * rtbhip.ET.Ry(qlim=[-1, 2])
    np.testing.assert_allclose(e._limits(False), [[-np.pi, 0, -1], [np.pi, 1, 2]])   # the device table's defaults, reference robot/ET.py:109-115
    with pytest.raises(ValueError):
        e.qlim                                                                          # the Python property: robot/ETS.py:335-337
    e.qlim = [[-1, 0, -1], [1, 0.5, 1]]
    assert e.qlim.shape == (2, 3)


def test_shard_range_partitions_exactly():
    for N, world in ((10, 4), (1000000, 8), (7, 8), (0, 3), (64, 1)):
        rows = [rtbhip.shard_range(N, r, world) for r in range(world)]
        assert rows[0][0] == 0
        assert sum(c for _, c in rows) == N
        for (b0, c0), (b1, _) in zip(rows, rows[1:]):
            assert b0 + c0 == b1
        assert max(c for _, c in rows) - min(c for _, c in rows) <= 1
    with pytest.raises(rtbhip.RtbHipError):
        rtbhip.shard_range(10, 4, 4)


def test_argument_shape_rules_match_reference():
    p = rtbhip.models.Panda().ets()
    q = np.arange(7.0)
    for qq in (q, list(q), q[None, :], q[:, None]):          # all ONE configuration (fknm.cpp:964-988)
        q2, single, tm = p._shape_q(qq)
        assert single and q2.shape == (1, 7) and not tm
    q2, single, _ = p._shape_q(np.zeros((5, 7)))
    assert not single and q2.shape == (5, 7)
    with pytest.raises(ValueError):
        p._shape_q(np.zeros((5, 6)))
    for bad in ("Wfgsrth", [object()] * 7, np.array(["a"] * 7)):   # symbolic => TypeError (test_ETS.py:363)
        with pytest.raises(TypeError):
            p._shape_q(bad)


@pytest.mark.skipif(rtbhip.device_count() > 0, reason="a GPU is present")
def test_compute_fails_loudly_without_gpu():
    p = rtbhip.models.Panda()
    with pytest.raises(rtbhip.RtbHipError):
        p.fkine(np.zeros(7))
    with pytest.raises(rtbhip.RtbHipError):
        p.jacob0(np.zeros((4, 7)))
    with pytest.raises(rtbhip.RtbHipError):
        rtbhip.models.DH.Puma560().rne(np.zeros(6), np.zeros(6), np.zeros(6))


def test_empty_batch_is_a_noop_everywhere():
    p = rtbhip.models.Panda().ets()
    assert p.eval(np.zeros((0, 7))).shape == (0, 4, 4)
    assert p.jacob0(np.zeros((0, 7))).shape == (0, 6, 7)
    pu = rtbhip.models.DH.Puma560()
    assert pu.rne(np.zeros((0, 6)), np.zeros((0, 6)), np.zeros((0, 6))).shape == (0, 6)


def test_dh_lowering_and_L24_match_independent_tables():
    from oracle import chains
    for prod, tab in ((rtbhip.models.DH.Puma560(), chains.puma560()), (rtbhip.models.DH.Panda(), chains.panda_dh())):
        np.testing.assert_array_equal(prod.L24(), tab.L24())
        rows = prod.ets().optable()
        ch = tab.ets()
        assert len(rows) == ch.m
        for i, (kind, flip, jindex, T) in enumerate(rows):
            assert kind == ch.kind[i] and flip == ch.flip[i]
            if kind == 6:
                np.testing.assert_allclose(np.asarray(T).reshape(16), ch.consts[i], atol=1e-16, rtol=0)
            else:
                assert jindex == ch.jindex[i]
        np.testing.assert_allclose(prod.qlim, tab.qlim.T)
    rows = rtbhip.models.Panda().ets().optable()
    ch = chains.panda_ets()
    for i, (kind, flip, jindex, T) in enumerate(rows):
        assert kind == ch.kind[i]
        if kind == 6:
            np.testing.assert_array_equal(np.asarray(T).reshape(16), ch.consts[i])


def test_robot_ets_start_end_follow_the_reference_path_rule():
    """BaseRobot.ets(start, end) (robot/BaseRobot.py:1555-1652 -> _find_ets :1426-1467): descending from `start` the path
    includes the start link's own transform; climbing multiplies by the inverse of every link left behind.  Checked
    numerically with the CPU oracle on a branched tree, and on the ETS-model facade's link0..linkK segments."""
    from oracle import oracle, chains
    from helpers import chain_from_ets
    ET, Link = rtbhip.ET, rtbhip.Link
    l0 = Link(ET.tz(0.3) * ET.Rz(), name="l0")
    l1 = Link(ET.tx(0.2) * ET.Ry(flip=True), name="l1", parent=l0)
    l2 = Link(ET.Rx(0.4) * ET.tz(), name="l2", parent=l1)
    r1 = Link(ET.ty(-0.1) * ET.Rx(), name="r1", parent=l0)
    r2 = Link(ET.SE3(chains.elementary("Rz", 0.5) @ chains.elementary("tx", 0.3)) * ET.Rz(), name="r2", parent=r1)
    rob = rtbhip.ERobot([l0, l1, l2, r1, r2])
    q = np.random.default_rng(0).uniform(-1, 1, rob.n)

    def fk(e):
        if e.n:                                 # chain_from_ets numbers the joints in order of appearance
            return oracle.fkine(chain_from_ets(e), q[e.jindices])[0]
        T = np.eye(4)
        for x in e:
            T = T @ x.T
        return T
    full = fk(rob.ets(end="l2"))
    nt.assert_allclose(fk(rob.ets()), fk(rob.ets(end=rob.links[-1])), atol=1e-15)
    # descending: start's own transform is part of the path
    nt.assert_allclose(fk(rob.ets(end="l0")) @ fk(rob.ets(start="l1", end="l2")), full, atol=1e-14)
    nt.assert_allclose(fk(rob.ets(start=l2, end=l2)), fk(rtbhip.ETS([e for e in rob._link_ets(l2)])), atol=1e-15)
    # across branches: up l2, l1 (inverted), down r1, r2
    cross = fk(rob.ets(start="l2", end="r2"))
    nt.assert_allclose(full @ cross, fk(rob.ets(end="r2")), atol=1e-13)
    assert [e.isflip for e in rob.ets(start="l2", end="l0") if e.isjoint][:2] == [True, False]     # tz -> flipped, Ry(flip) -> unflipped
    with pytest.raises(ValueError):
        rob.ets(end="nope")
    with pytest.raises(ValueError):
        rob.ets(start=Link(ET.Rz(), name="stranger"))
    # ETS-model facade: the reference's Robot(ETS) link segments
    panda = rtbhip.models.Panda()
    segs = panda.ets().split()
    assert len(segs) == 8 and all(s[-1].isjoint for s in segs[:7]) and not any(e.isjoint for e in segs[7])
    assert panda.ets() is panda.ets(None, None)
    part = panda.ets(start="link2", end="link4")
    assert part.n == 3 and len(part) == sum(len(s) for s in segs[2:5])
    assert panda.ets(end=3).n == 4
    with pytest.raises(ValueError):
        panda.ets(start="link5", end="link2")
    e = ET.Rz(jindex=0) * ET.tx(0.3) * ET.SE3(chains.elementary("Ry", 0.2))
    inv = e.inv()
    assert [x.axis for x in inv] == ["SE3", "tx", "Rz"] and inv[2].isflip and inv[1].eta == -0.3
    nt.assert_allclose(inv[0].T @ e[2].T, np.eye(4), atol=1e-15)

"""fkine_all: the poses of every link frame (reference DHRobot.fkine_all robot/DHRobot.py:1012-1064, Robot.fkine_all
robot/Robot.py:638-698) from one chain walk per configuration.  Pins: the eight frames of the DH Panda at q = 1..7
(reference tests/test_DHRobot.py:638-710, 4 decimals)."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import urdf
from oracle import oracle, chains
from helpers import literals, chain_from_ets, product_ets, mixed_spec, tool_base

LIT = literals()


def _dh_marks(robot):
    k = 1 if (robot.base is not None and not np.array_equal(robot.base, np.eye(4))) else 0
    marks = [k]
    for l in robot.links:
        k += len(l.ets())
        marks.append(k)
    return marks


def test_oracle_frames_G11_dh_panda():
    panda = rtbhip.models.DH.Panda()
    ch = chain_from_ets(panda.ets())
    q = np.arange(1, 8, dtype=float)
    F = oracle.link_frames(ch, q, _dh_marks(panda))[0]
    nt.assert_array_almost_equal(F[0], np.eye(4), decimal=12)                      # t0
    for k in range(1, 8):
        nt.assert_array_almost_equal(F[k], LIT["G11_dh_panda_t%d" % k], decimal=4)
    # the last frame times the tool is fkine
    nt.assert_allclose(F[7] @ panda.tool, oracle.fkine(ch, q)[0], atol=1e-12)


def test_emu_frames_every_mark_of_the_mixed_chain():
    """Marks after every transform of a chain with all six joint kinds, flips and an SE3 constant: the residual constant
    run (incl. the axis permutation an x / y joint leaves behind) must be applied at each of them."""
    import emu_harness as emu
    spec = mixed_spec()
    ets, ch = product_ets(spec), chains.Chain(spec, name="mixed")
    rng = np.random.default_rng(2)
    q = rng.uniform(-2, 2, (5, ch.n))
    tool, base = tool_base()
    for marks in (list(range(ch.m + 1)), [0, 0, 3, 3, ch.m], [ch.m], []):
        for b in (None, base):
            got = emu.link_frames(ets, q, marks, base=b)
            nt.assert_allclose(got, oracle.link_frames(ch, q, marks, base=b), atol=1e-12)
    panda = rtbhip.models.DH.Panda()
    F = emu.link_frames(panda.ets(), np.arange(1, 8.0), _dh_marks(panda))[0]
    for k in range(1, 8):
        nt.assert_array_almost_equal(F[k], LIT["G11_dh_panda_t%d" % k], decimal=4)


@pytest.mark.gpu
def test_gpu_fkine_all_dh_robots():
    """Host buffers only (the CPU replay of the suite runs this too)."""
    panda = rtbhip.models.DH.Panda()
    q1 = np.arange(1, 8, dtype=float)
    F = panda.fkine_all(q1)                                                           # reference tests/test_DHRobot.py:638-710
    assert F.shape == (8, 4, 4)
    nt.assert_array_almost_equal(F[0], np.eye(4), decimal=12)
    for k in range(1, 8):
        nt.assert_array_almost_equal(F[k], LIT["G11_dh_panda_t%d" % k], decimal=4)
    rng = np.random.default_rng(4)
    for robot in (panda, rtbhip.models.DH.Puma560()):
        ch = chain_from_ets(robot.ets())
        N = 1000
        q = rng.uniform(-3, 3, (N, robot.n))
        F = robot.fkine_all(q)
        assert F.shape == (N, robot.n + 1, 4, 4)
        nt.assert_allclose(F[:200], oracle.link_frames(ch, q[:200], _dh_marks(robot)), atol=1e-10)
        last = F[:, -1] if robot.tool is None else F[:, -1] @ robot.tool
        nt.assert_allclose(last, robot.fkine(q), atol=1e-12)
    # marks: errors and the empty cases
    e = panda.ets()
    with pytest.raises(rtbhip.RtbHipError):
        e.link_frames(q1, [3, 2])
    with pytest.raises(rtbhip.RtbHipError):
        e.link_frames(q1, [len(e) + 1])
    assert e.link_frames(np.zeros((0, 7)), [0, 1]).shape == (0, 2, 4, 4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["YuMi", "UR5", "Fetch", "Panda"])
def test_gpu_fkine_all_urdf_trees_index_by_link_number(name):
    """Branched URDF robots: slot 0 is the base, slot `link.number` is that link's frame -- the reference's contract
    (robot/Robot.py:679 writes Tall[link.number]); `er.links` is in sorted order, `number` is the place in the list the robot was built from, so
    the two differ for every branched robot.  Each frame equals fkine of the path to that link.  Host buffers only."""
    rng = np.random.default_rng(5)
    u = urdf.load(name)
    er = u.erobot()
    q = rng.uniform(-1, 1, (50, er.n))
    F = er.fkine_all(q)
    assert F.shape == (50, len(er.links) + 1, 4, 4)
    nt.assert_allclose(F[:, 0], np.broadcast_to(np.eye(4), (50, 4, 4)), atol=0)
    assert sorted(l.number for l in er.links) == list(range(1, len(er.links) + 1))
    for l in er.links:
        e = er.ets(end=l)
        want = e.eval(q[:, :e.q_width]) if e.n else np.broadcast_to(e.eval(np.zeros(0)), (50, 4, 4))
        nt.assert_allclose(F[:, l.number], want, atol=1e-12)
    nt.assert_array_equal(u.fkine_all(q), F)
    one = er.fkine_all(q[3])
    nt.assert_array_equal(one, F[3])


@pytest.mark.gpu
def test_gpu_fkine_all_device_tensors():
    """The same calls on device tensors give the host-buffer results bit for bit (needs a device)."""
    import torch
    rng = np.random.default_rng(6)
    for robot in (rtbhip.models.DH.Panda(), rtbhip.models.DH.Puma560()):
        q = rng.uniform(-3, 3, (1000, robot.n))
        Ft = robot.fkine_all(torch.from_numpy(q).cuda())
        nt.assert_array_equal(Ft.cpu().numpy(), robot.fkine_all(q))
    er = urdf.load("YuMi").erobot()
    q = rng.uniform(-1, 1, (50, er.n))
    Ft = er.fkine_all(torch.from_numpy(q).cuda())
    nt.assert_array_equal(Ft.cpu().numpy(), er.fkine_all(q))

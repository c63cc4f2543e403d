"""Joint axes that are not a coordinate axis: the reference's lowering (tools/urdf/urdf.py:1357-1369, 1706-1722) restated statement by statement.

  * the axis is NORMALISED when the <axis> tag is parsed (:1365-1368);
  * an axis with two or more nonzero components is "normalised onto z": `u, n = unitvec_norm(v); R = angvec2r(n, u)` -- a rotation ABOUT the
    axis by its norm, which after the parse-time normalisation is always 1 radian, whatever length the file wrote -- post-multiplied onto
    RPY(rpy), taken through tr2rpy and back, and the joint then turns about / slides along z (:1710-1722);
  * a coordinate axis of any length is that coordinate axis (flip for the negative direction).

None of the 20 model files the reference ships has a skew axis (they pin the rest of the lowering, tests/test_xacro.py); rtbhip kept the written
length until round 4 (found by a fuzz run: `1 1 0` was rotated by 1.414 rad).  Host-side: no GPU needed."""
import math

import numpy as np
import numpy.testing as nt

from oracle import sm_standin
from rtbhip import urdf


def reference_constant(xyz, rpy, axis_as_written):
    sm, smb = sm_standin.modules()
    v = np.asarray(axis_as_written, dtype=np.float64)
    norm = np.linalg.norm(v)
    if norm != 0:
        v = v / norm                                              # Joint.axis setter, :1365-1368
    if np.count_nonzero(v) < 2:                                   # :1706-1707
        return (sm.SE3(*xyz) * sm.SE3.RPY(rpy)).A, v
    n = np.linalg.norm(v); u = v / n                              # unitvec_norm
    x, y, z = u
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]])
    R = np.eye(3) + math.sin(n) * K + (1 - math.cos(n)) * (K @ K)  # angvec2r(n, u)
    Rt = np.eye(4); Rt[:3, :3] = sm.SE3.RPY(rpy).A[:3, :3] @ R     # SE3.RPY(joint.rpy) * R
    rpy2 = smb.tr2rpy(Rt)                                          # :1717
    return (sm.SE3(*xyz) * sm.SE3.RPY(rpy2)).A, np.array([0.0, 0.0, 1.0])


def joint_xml(typ, xyz, rpy, axis):
    return ('<robot name="r"><link name="a"/><link name="b"/><joint name="j" type="%s"><parent link="a"/><child link="b"/>'
            '<origin xyz="%r %r %r" rpy="%r %r %r"/><axis xyz="%r %r %r"/><limit lower="-2" upper="2" effort="1" velocity="1"/></joint></robot>'
            % ((typ,) + tuple(map(float, xyz)) + tuple(map(float, rpy)) + tuple(map(float, axis))))


def test_skew_and_scaled_axes_are_lowered_as_the_reference_lowers_them():
    rng = np.random.default_rng(5)
    cases = [([1, 1, 0], 1.0), ([1, 1, 0], 3.7), ([0.3, 0.4, 0.8660254037844386], 1.0), ([0, 0, 3.7], 1.0), ([0, -0.2, 0], 1.0), ([1e-3, 0, 1], 1.0)]
    cases += [(rng.normal(size=3), float(rng.uniform(0.2, 5.0))) for _ in range(20)]
    for axis, scale in cases:
        axis = np.asarray(axis, dtype=np.float64) * scale
        xyz, rpy = rng.uniform(-0.3, 0.3, 3), rng.uniform(-3, 3, 3)
        for typ in ("revolute", "prismatic"):
            r = urdf.loadstr(joint_xml(typ, xyz, rpy, axis))
            j = r.joints[0]
            want, ax = reference_constant(xyz, rpy, axis)
            nt.assert_allclose(j.constant(), want, atol=1e-14, err_msg=str(axis))
            var = j.variable()
            k = int(np.argmax(np.abs(ax)))
            assert var.axis == (("Rx", "Ry", "Rz")[k] if typ == "revolute" else ("tx", "ty", "tz")[k]) and var.isflip == bool(ax[k] < 0)
    # the length written for a skew axis does not matter (it did until round 4)
    a = urdf.loadstr(joint_xml("revolute", [0, 0, 0], [0, 0, 0], [1, 1, 0])).joints[0].constant()
    b = urdf.loadstr(joint_xml("revolute", [0, 0, 0], [0, 0, 0], [5, 5, 0])).joints[0].constant()
    nt.assert_allclose(a, b, atol=1e-15)


def test_a_fixed_joint_with_a_skew_axis_tag_gets_the_rotation_too():
    """the reference tests the axis before it looks at the joint's type (:1706 against :1724-1755): a fixed joint that carries a skew <axis> tag has the
    normalising rotation in its constant transform."""
    xyz, rpy, axis = [0.1, -0.2, 0.3], [0.4, -0.5, 0.6], [1.0, 2.0, 0.0]
    j = urdf.loadstr(joint_xml("fixed", xyz, rpy, axis).replace('<limit lower="-2" upper="2" effort="1" velocity="1"/>', "")).joints[0]
    nt.assert_allclose(j.constant(), reference_constant(xyz, rpy, axis)[0], atol=1e-14)
    assert j.variable() is None


def test_joint_numbers_of_a_branched_robot_are_the_references_depth_first_ones():
    """A file that lists one branch's tail after the other branch (YuMi's grippers): the robot-wide joint numbers are the reference's -- depth first
    from the base link (BaseRobot._sort_links) -- in URDFRobot.jindex, in ets(compact=False) and in the ERobot made from it (the dynamics), so
    that fkine and rne of one object read the same column for the same joint.  Until round 4 URDFRobot numbered in FILE order."""
    def joint(name, parent, child, axis):
        return ('<joint name="%s" type="revolute"><parent link="%s"/><child link="%s"/><origin xyz="0.1 0 0"/><axis xyz="%s"/>'
                '<limit lower="-2" upper="2" effort="1" velocity="1"/></joint>' % (name, parent, child, axis))
    links = "".join('<link name="%s"/>' % n for n in ("base", "a1", "a2", "b1", "b2", "a3"))
    xml = '<robot name="y">' + links + joint("ja1", "base", "a1", "0 0 1") + joint("ja2", "a1", "a2", "0 1 0") + joint("jb1", "base", "b1", "0 0 1") \
        + joint("jb2", "b1", "b2", "0 1 0") + joint("ja3", "a2", "a3", "1 0 0") + "</robot>"            # ja3 (branch a's tail) comes last in the file
    r = urdf.loadstr(xml)
    assert r.jindex == {"ja1": 0, "ja2": 1, "ja3": 2, "jb1": 3, "jb2": 4}
    er = r.erobot()
    assert {l.name: l.jindex for l in er.links if l.isjoint} == {"a1": 0, "a2": 1, "a3": 2, "b1": 3, "b2": 4}
    assert list(r.ets(end="a3", compact=False).jindices) == [0, 1, 2] and list(r.ets(end="b2", compact=False).jindices) == [3, 4]
    yumi = urdf.load("YuMi")
    assert {l.joint.name: l.jindex for l in []} == {}                                      # (erobot links carry the link name; compare through it)
    by_link = {l.name: l.jindex for l in yumi.erobot().links if l.isjoint}
    assert {j.child: yumi.jindex[j.name] for j in yumi.joints if j.name in yumi.jindex} == by_link

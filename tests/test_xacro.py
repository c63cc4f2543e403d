"""rtbhip.xacro -- the xacro expander in front of Robot.URDF (the reference reads its robot models from xacro files,
robot/Robot.py:218-286).  Host-side text processing: none of these tests needs a GPU.

  * the synthetic description of tests/golden/xacro (written for this repository; every feature the expander implements) against
    tests/golden/xacro/demo_arm.expected.urdf -- what the REFERENCE's own xacro tool makes of the same files (tests/golden/make_golden.py xacro);
  * where /root/reference exists (the build container): EVERY xacro file the reference's model classes read (models/URDF/*.py: 20 files, PR2's 92
    links included) expanded by rtbhip.xacro and by the reference's tool, compared element for element, attribute string for attribute string;
  * arguments, package look-up, optional includes, and the refusals.
"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import rtbhip
from rtbhip import urdf, xacro

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xacro")
DEMO = os.path.join(HERE, "demo_description", "urdf", "demo_arm.urdf.xacro")
REF_DATA = "/root/reference/rtb-data/rtbdata/xacro"
REF_TOOLS = "/root/reference/src/roboticstoolbox/tools"


def canon(e):
    return (e.tag, tuple(sorted(e.attrib.items())), (e.text or "").strip(), tuple(canon(c) for c in e))


def first_difference(a, b, path=""):
    if a[:3] != b[:3]:
        return "%s: %r != %r" % (path, a[:3], b[:3])
    if len(a[3]) != len(b[3]):
        return "%s: %d children != %d" % (path, len(a[3]), len(b[3]))
    for c, d in zip(a[3], b[3]):
        r = first_difference(c, d, path + "/" + c[0])
        if r:
            return r
    return None


def test_synthetic_description_equals_what_the_reference_tool_makes_of_it():
    mine = canon(ET.fromstring(xacro.process(DEMO)))
    want = canon(ET.fromstring(open(os.path.join(HERE, "demo_arm.expected.urdf")).read()))
    assert first_difference(want, mine) is None
    # and it is a robot the loader accepts: 5 joints, 4 of them actuated, limits and inertials through
    u = urdf.read(DEMO)
    assert [l.name for l in u.links] == ["d_base", "d_shoulder", "d_elbow", "d_wrist_a", "d_wrist_b", "d_tool"] and u.n == 4
    e = u.ets(end="d_tool")
    np.testing.assert_allclose(e.qlim[:, 0], [-np.radians(170), np.radians(170)])
    np.testing.assert_allclose(e.qlim[:, 1], [0.0, 0.4])
    assert u.linkdict["d_elbow"].m == 2.0 and u.linkdict["d_shoulder"].m == 0.75


def test_arguments_change_the_robot():
    text = xacro.process(DEMO, mappings={"prefix": "x_", "with_tool": "false", "reach": "3"})
    root = ET.fromstring(text)
    names = [l.get("name") for l in root.findall("link")]
    assert names == ["x_base", "x_shoulder", "x_elbow", "x_wrist_a", "x_wrist_b"]            # no tool
    assert root.find("meta") is None                                                          # upper = 3 > 1: the `unless` block is gone
    lim = root.findall("joint")[1].find("limit")
    assert float(lim.get("upper")) == 3.0
    r = rtbhip.ERobot.URDF(DEMO)                                                               # Robot.URDF on an xacro file: defaults
    assert r.n == 4 and len(r.links) == 6
    u = urdf.read(DEMO, mappings={"with_tool": "false"})
    assert len(u.links) == 5


def test_packages_and_optional_includes():
    f = os.path.join(HERE, "demo_description", "urdf", "uses_support.urdf.xacro")
    for packages in (None, [HERE], {"demo_support": os.path.join(HERE, "demo_support")}):     # by ancestry, by search root, by name
        root = ET.fromstring(xacro.process(f, packages=packages))
        sizes = [l.find("visual/geometry/box").get("size") for l in root.findall("link")]
        assert sizes == ["0.5 0.5 0.5", "1 1 1"]                                               # the macro's default is evaluated per call


def _write(tmp_path, body, name="t.xacro"):
    p = tmp_path / name
    p.write_text('<robot name="t" xmlns:xacro="http://www.ros.org/wiki/xacro">%s</robot>' % body)
    return str(p)


def test_language_details(tmp_path):
    def run(body, **kw):
        return ET.fromstring(xacro.process(_write(tmp_path, body), **kw))
    # lazy properties, types, division, a conditional expression, string concatenation
    r = run('<xacro:property name="b" value="${a + 1}"/><xacro:property name="a" value="2"/><x v="${b}" w="${a / 4}" s="p${a}q" t="${\'yes\' if b == 3 else \'no\'}"/>')
    assert r.find("x").attrib == {"v": "3", "w": "0.5", "s": "p2q", "t": "yes"}
    # scopes: a macro's properties stay inside it unless sent out; parameters shadow globals
    r = run('<xacro:property name="g" value="1"/><xacro:macro name="m" params="g:=5"><xacro:property name="loc" value="9"/>'
            '<xacro:property name="up" value="${g * 2}" scope="parent"/><y g="${g}" loc="${loc}"/></xacro:macro><xacro:m/><z g="${g}" up="${up}"/>')
    assert r.find("y").attrib == {"g": "5", "loc": "9"} and r.find("z").attrib == {"g": "1", "up": "10"}
    # forwarded parameters, with and without a fallback
    r = run('<xacro:macro name="inner" params="k:=^ j:=^|4"><i k="${k}" j="${j}"/></xacro:macro>'
            '<xacro:macro name="outer" params="k"><xacro:inner/></xacro:macro><xacro:outer k="8"/>')
    assert r.find("i").attrib == {"k": "8", "j": "4"}
    # *block inserts the element, **block what it contains; blocks are evaluated where they were written
    r = run('<xacro:property name="v" value="7"/><xacro:macro name="m" params="v *one **many"><o><xacro:insert_block name="one"/></o>'
            '<n><xacro:insert_block name="many"/></n></xacro:macro><xacro:m v="0"><a v="${v}"/><wrap><b/><c/></wrap></xacro:m>')
    assert r.find("o/a").get("v") == "7" and [c.tag for c in r.find("n")] == ["b", "c"]
    # if / unless accept booleans, numbers and their spellings; text nodes are evaluated too
    r = run('<xacro:if value="1"><a/></xacro:if><xacro:unless value="false"><b/></xacro:unless><xacro:if value="${2 &lt; 1}"><c/></xacro:if><t>${3 * 2} $${kept}</t>')
    assert [c.tag for c in r] == ["a", "b", "t"] and r.find("t").text == "6 ${kept}"
    # $(arg), $(optenv), $(eval)
    os.environ.pop("RTBHIP_XACRO_TEST", None)
    r = run('<xacro:arg name="n" default="3"/><a n="$(arg n)" e="$(optenv RTBHIP_XACRO_TEST fallback value)" v="$(eval 2 ** 5)" m="${arg(\'n\') + 1}"/>', mappings={"n": "4"})
    assert r.find("a").attrib == {"n": "4", "e": "fallback value", "v": "32", "m": "5"}


@pytest.mark.parametrize("body,fragment", [
    ('<xacro:nosuch/>', "unknown macro"),
    ('<xacro:macro name="m" params="a"/><xacro:m/>', "parameter 'a' is missing"),
    ('<xacro:macro name="m" params="a"/><xacro:m a="1" b="2"/>', "unknown parameter"),
    ('<xacro:macro name="m" params="*blk"/><xacro:m/>', "block parameter"),
    ('<a v="${undefined_name}"/>', "undefined_name"),
    ('<xacro:property name="p" value="${p}"/><a v="${p}"/>', "in terms of itself"),
    ('<xacro:if value="maybe"><a/></xacro:if>', "not a boolean"),
    ('<xacro:include filename="missing.xacro"/>', "no such file"),
    ('<a v="$(arg nothing)"/>', "undefined substitution argument"),
    ('<xacro:element xacro:name="x"/>', "not supported"),
    ('<a v="${load_yaml(1)}"/>', "load_yaml"),
    ('<a v="${__import__(\'os\')}"/>', "double underscores"),
    ('<a v="${().__class__.__base__.__subclasses__()}"/>', "double underscores"),
    ('<a v="${().\uff3f\uff3fclass\uff3f\uff3f}"/>', "double underscores"),          # fullwidth low lines: Python NFKC-normalises identifiers
    ('<a v="${()._x}"/>', "starting with an underscore"),
    ('<a v="${open(\'/etc/passwd\').read()}"/>', "open"),
])
def test_refusals(tmp_path, body, fragment):
    with pytest.raises(xacro.XacroError) as e:
        xacro.process(_write(tmp_path, body))
    assert fragment in str(e.value)


REF_FILES = ["franka_description/robots/frankie_arm_hand.urdf.xacro", "franka_description/robots/panda_arm_hand.urdf.xacro",
             "kinova_description/urdf/j2n4s300_standalone.xacro", "kortex_description/robots/gen3.xacro",
             "kuka_description/kuka_lbr_iiwa/urdf/lbr_iiwa_14_r820.xacro", "puma560_description/urdf/puma560_robot.urdf.xacro",
             "ridgeback_description/urdf/ridgeback.urdf.xacro", "pr2_description/robots/pr2.urdf.xacro",
             "ur_description/urdf/ur3_joint_limited_robot.urdf.xacro", "ur_description/urdf/ur5_joint_limited_robot.urdf.xacro",
             "ur_description/urdf/ur10_joint_limited_robot.urdf.xacro"] + \
            ["interbotix_descriptions/urdf/%s.urdf.xacro" % k for k in ("px100", "px150", "rx150", "rx200", "vx300", "vx300s", "wx200", "wx250", "wx250s")]


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference's data package is only in the build container")
def test_every_xacro_file_the_reference_models_read_expands_as_the_reference_tool_expands_it():
    from pathlib import PurePosixPath
    if REF_TOOLS not in sys.path:
        sys.path.append(REF_TOOLS)              # appended, not prepended: the folder also holds a types.py
    import xacro as ref_tool
    for f in REF_FILES:
        p = PurePosixPath(REF_DATA) / f
        tld = PurePosixPath(REF_DATA) / "pr2_description" if f.startswith("pr2") else None      # models/URDF/PR2.py passes xacro_tld
        want = canon(ET.fromstring(ref_tool.main(p, tld)))
        mine = canon(ET.fromstring(xacro.process(str(p))))
        assert first_difference(want, mine) is None, f


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference's data package is only in the build container")
def test_fleet_arms_straight_from_the_reference_xacro_data_equal_the_shipped_descriptions():
    """BASELINE config 5's arms come 'straight from rtb-data/rtbdata/xacro/**' (SURVEY 8f-3): read at run time through rtbhip.xacro they are
    the robots the shipped (pre-expanded) descriptions hold -- links, joints, constant transforms, axes, limits, inertials."""
    for name in urdf.FLEET16:
        rel = [k for k, v in urdf.REFERENCE_PATHS.items() if v == name][0]
        live = urdf.read(os.path.join(REF_DATA, rel))
        kept = urdf.load(name)
        assert [l.name for l in live.links] == [l.name for l in kept.links] and [j.name for j in live.joints] == [j.name for j in kept.joints]
        for a, b in zip(live.joints, kept.joints):
            assert (a.type, a.parent, a.child) == (b.type, b.parent, b.child)
            np.testing.assert_array_equal(a.constant(), b.constant())
        for a, b in zip(live.links, kept.links):
            assert a.m == b.m and np.array_equal(np.asarray(a.r), np.asarray(b.r)) and np.array_equal(np.asarray(a.I), np.asarray(b.I))
        ea, eb = live.ets(end=kept.ee), kept.ets(end=kept.ee, start=None) if name not in urdf._MODEL_EE else None
        if eb is not None:
            np.testing.assert_array_equal(ea.qlim, eb.qlim)

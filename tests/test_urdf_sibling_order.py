"""Round-4 advisor findings, as tests.

1. Sibling links declared in the OPPOSITE order to their joints: the robot-wide joint numbering walks children in LINK file order
   (reference BaseRobot._sort_links robot/BaseRobot.py:264-267 appends `link.parent._children` while walking the link list that
   tools/urdf/urdf.py:1666-1676 built in link order), so URDFRobot (fkine / jacob0 with the robot-wide q) and the ERobot made from it
   (rne / inertia / fkine_all) read the SAME column for the same joint.
2. `robot.qlim = X` must reach the chains `robot.ets()` builds (the limits ik_LM draws restarts from and rejects by): in the reference
   Link.qlim writes through to the joint's ET (robot/Link.py:1010-1040).
"""
import numpy as np

import cpu_backend
from rtbhip import urdf

XML = """<robot name="sib">
<link name="base"/><link name="R"/><link name="L"/><link name="R2"/><link name="L2"/>
<joint name="jL" type="revolute"><parent link="base"/><child link="L"/><origin xyz="0 0.1 0"/><axis xyz="0 0 1"/><limit lower="-1" upper="1" effort="1" velocity="1"/></joint>
<joint name="jR" type="revolute"><parent link="base"/><child link="R"/><origin xyz="0 -0.1 0"/><axis xyz="0 1 0"/><limit lower="-2" upper="2" effort="1" velocity="1"/></joint>
<joint name="jL2" type="revolute"><parent link="L"/><child link="L2"/><origin xyz="0.2 0 0"/><axis xyz="1 0 0"/><limit lower="-1.5" upper="1.5" effort="1" velocity="1"/></joint>
<joint name="jR2" type="prismatic"><parent link="R"/><child link="R2"/><origin xyz="0.3 0 0"/><axis xyz="0 0 1"/><limit lower="0" upper="0.5" effort="1" velocity="1"/></joint>
</robot>"""


def test_sibling_links_declared_opposite_to_their_joints_share_one_numbering():
    r = urdf.loadstr(XML)
    # links in file order: base, R, L, R2, L2 -> depth first with siblings in LINK order: R (0), R2 (1), L (2), L2 (3)
    assert r.jindex == {"jR": 0, "jR2": 1, "jL": 2, "jL2": 3}
    e = r.erobot()
    number = {l.name: l.jindex for l in e.links}
    assert {j.child: r.jindex[j.name] for j in r.joints} == {k: v for k, v in number.items() if v is not None}
    with cpu_backend.installed():
        rng = np.random.default_rng(0)
        q = rng.uniform(-0.4, 0.4, (3, 4))
        q[:, 1] = np.abs(q[:, 1])
        for end in ("R2", "L2"):
            a = np.asarray(r.fkine(q, end=end)).reshape(-1, 4, 4)                 # URDFRobot: robot-wide q by its jindex
            b = np.asarray(e.ets(end=end).eval(q)).reshape(-1, 4, 4)              # ERobot: the reference's numbering
            np.testing.assert_allclose(a, b, atol=1e-14)


def test_robot_qlim_assignment_reaches_the_chains():
    e = urdf.loadstr(XML).erobot()
    old = e.ets(end="L2").qlim.copy()
    new = np.array([[-0.3, 0.0, -0.2, -0.1], [0.3, 0.25, 0.2, 0.1]])
    e.qlim = new
    np.testing.assert_array_equal(e.qlim, new)
    got = e.ets(end="L2").qlim                                                     # joints jL (column 2) and jL2 (column 3) on this path
    assert not np.array_equal(got, old)
    lim = {int(et.jindex): np.asarray(et.qlim).reshape(2) for et in e.ets(end="L2") if et.isjoint}
    np.testing.assert_array_equal(lim[2], new[:, 2])
    np.testing.assert_array_equal(lim[3], new[:, 3])
    lim = {int(et.jindex): np.asarray(et.qlim).reshape(2) for et in e.ets(end="R2") if et.isjoint}
    np.testing.assert_array_equal(lim[0], new[:, 0])
    np.testing.assert_array_equal(lim[1], new[:, 1])

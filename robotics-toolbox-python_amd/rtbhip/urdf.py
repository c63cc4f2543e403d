"""URDF -> elementary-transform-sequence lowering for the GPU kinematics path.

What it replaces in the reference: ``tools/urdf/urdf.py:1662-1760`` (``URDF.__init__``: one
``ET.SE3(trans * RPY) [* joint ET]`` per joint), ``Robot.URDF_read`` (``robot/Robot.py:218-284``) and
the serial-path part of ``BaseRobot.ets(start, end)`` (``robot/BaseRobot.py:1426-1467``) -- without
spatialmath, and only for what the batched kernels need: joint origins, axes, limits, the link tree,
and each link's ``<inertial>`` block (kept for the dynamics rows of SURVEY 8f).

Only plain URDF is parsed.  The robot descriptions the reference ships as xacro
(rtb-data/rtbdata/xacro) are pre-expanded to kinematic URDFs in ``rtbhip/data/urdf`` by
``scripts/make_urdf_data.py``; ``load("Panda")`` etc. read those.

Lowering rules (same as the reference):
  * constant part of a joint = Trans(origin xyz) * RPY(origin rpy), RPY in the URDF/'zyx' order
    R = Rz(yaw) Ry(pitch) Rx(roll)                                   (urdf.py:1704-1709)
  * axis aligned with +-x/+-y/+-z -> Rx/Ry/Rz (revolute, continuous) or tx/ty/tz (prismatic), with
    ``flip`` for the negative direction                               (urdf.py:1726-1754)
  * any other axis: the constant part is post-multiplied by the reference's normalising rotation
    ``angvec2r(|v|, v/|v|)`` and the joint then acts about/along z   (urdf.py:1710-1722)
  * fixed (and unsupported floating/planar) joints contribute the constant part only.
"""
import math
import os
import xml.etree.ElementTree as XT

import numpy as np

from .et import ET, ETS, _poses

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "urdf")


def _floats(text, n, default):
    if text is None:
        return np.array(default, dtype=np.float64)
    v = np.array([float(x) for x in text.split()], dtype=np.float64)
    if v.size != n:
        raise ValueError("expected %d numbers, got %r" % (n, text))
    return v


def rpy_matrix(rpy):
    """R = Rz(yaw) Ry(pitch) Rx(roll) -- spatialmath SE3.RPY(order='zyx'), the URDF convention."""
    r, p, y = (float(v) for v in rpy)
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], dtype=np.float64)
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]], dtype=np.float64)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=np.float64)
    return Rz @ Ry @ Rx


def angvec_matrix(theta, v):
    """Rodrigues rotation by theta about unit vector v (spatialmath angvec2r)."""
    x, y, z = v
    K = np.array([[0, -z, y], [z, 0, -x], [-y, x, 0]], dtype=np.float64)
    return np.eye(3) + math.sin(theta) * K + (1 - math.cos(theta)) * (K @ K)


class URDFJoint:
    def __init__(self, el):
        self.name = el.get("name")
        self.type = el.get("type")
        self.parent = el.find("parent").get("link")
        self.child = el.find("child").get("link")
        o = el.find("origin")
        self.xyz = _floats(None if o is None else o.get("xyz"), 3, [0, 0, 0])
        self.rpy = _floats(None if o is None else o.get("rpy"), 3, [0, 0, 0])
        a = el.find("axis")
        self.axis = _floats(None if a is None else a.get("xyz"), 3, [1, 0, 0])   # URDF default axis
        norm = float(np.linalg.norm(self.axis))
        if norm != 0:
            self.axis = self.axis / norm          # the reference normalises when it parses (tools/urdf/urdf.py:1357-1369): its "normalising rotation" of
                                                  # a skew axis (constant() below) is therefore always by |v/|v|| = 1 radian, whatever the length written
        # limits as the reference's reader takes them (tools/urdf/urdf.py:1391-1402, 1762-1768): a revolute or prismatic joint MUST carry a
        # <limit> (ValueError otherwise); whatever joint has one -- a `continuous` joint too: the Mico's and the Gen3's are limited to +-2 pi this
        # way -- gets qlim = [lower, upper].  An attribute that is missing leaves a None in the reference's array (robot.qlim then reads a
        # revolute joint as [-pi, pi] and a prismatic one as NaN): here such a half-written limit is simply no limit.
        lim = el.find("limit")
        self.lower = self.upper = None
        if lim is None and self.type in ("revolute", "prismatic"):
            raise ValueError("Require joint limit for prismatic and revolute joints")
        if lim is not None and lim.get("lower") is not None and lim.get("upper") is not None:
            self.lower = float(lim.get("lower"))
            self.upper = float(lim.get("upper"))
        elif lim is not None and (lim.get("lower") is not None or lim.get("upper") is not None):
            # half a limit: the reference stores [lower, None] / [None, upper] and robot.qlim (robot/BaseRobot.py:1017-1030) then reads a revolute
            # joint whose lower bound is missing as [-pi, pi] (a missing UPPER bound has no defined reading there).  Same outcome here -- no limit,
            # i.e. [-pi, pi] for a revolute joint -- but not silently: the URDF specification's own default for a missing bound would be 0
            import warnings
            warnings.warn("URDF joint %r: <limit> gives only one of lower / upper -- treated as no joint limit (the reference reads a revolute "
                          "joint without a lower bound as [-pi, pi]; the URDF default for a missing bound is 0)" % self.name, stacklevel=3)
        d = el.find("dynamics")
        self.friction = None if d is None or d.get("friction") is None else float(d.get("friction"))
        self.damping = None if d is None or d.get("damping") is None else float(d.get("damping"))

    @property
    def actuated(self):
        return self.type in ("revolute", "continuous", "prismatic")

    def constant(self):
        """4x4 constant part, incl. the reference's normalising rotation for a skew axis."""
        T = np.eye(4)
        R = rpy_matrix(self.rpy)
        if np.count_nonzero(self.axis) >= 2:       # (whatever the joint's type: the reference tests the axis before it looks at the type, :1706)
            n = float(np.linalg.norm(self.axis))
            R = R @ angvec_matrix(n, self.axis / n)                  # urdf.py:1713-1718
        T[:3, :3] = R
        T[:3, 3] = self.xyz
        return T

    def variable(self, jindex=None):
        """The joint's variable ET (None for a fixed joint)."""
        if not self.actuated:
            return None
        ax = self.axis
        if np.count_nonzero(ax) >= 2:
            k, flip = 2, False                                        # urdf.py:1722
        else:
            k = int(np.argmax(np.abs(ax)))
            if ax[k] == 0:
                return None
            flip = bool(ax[k] < 0)
        rot = self.type in ("revolute", "continuous")
        name = ("Rx", "Ry", "Rz")[k] if rot else ("tx", "ty", "tz")[k]
        qlim = None if self.lower is None else [self.lower, self.upper]
        return ET(name, flip=flip, jindex=jindex, qlim=qlim)


class URDFLink:
    def __init__(self, el):
        self.name = el.get("name")
        self.m = 0.0
        self.r = np.zeros(3)
        self.I = np.zeros((3, 3))
        self.parent = None      # URDFLink
        self.joint = None       # URDFJoint that attaches this link to its parent
        self.children = []
        ine = el.find("inertial")
        if ine is not None:
            mass = ine.find("mass")
            if mass is not None:
                self.m = float(mass.get("value", 0.0))
            o = ine.find("origin")
            if o is not None:
                self.r = _floats(o.get("xyz"), 3, [0, 0, 0])
            I = ine.find("inertia")
            if I is not None:
                g = lambda k: float(I.get(k, 0.0))
                self.I = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")],
                                   [g("ixz"), g("iyz"), g("izz")]])


class URDFRobot:
    """Link tree of one URDF with batched kinematics over any root->link path."""

    def __init__(self, urdf_string, name=None, ee=None, tool=None):
        root = XT.fromstring(urdf_string)
        if root.tag != "robot":
            raise ValueError("not a URDF: root element is <%s>" % root.tag)
        self.name = name or root.get("name", "")
        self.urdf_string = urdf_string
        self.links = [URDFLink(e) for e in root.findall("link")]
        self.joints = [URDFJoint(e) for e in root.findall("joint")]
        self.linkdict = {l.name: l for l in self.links}
        if len(self.linkdict) != len(self.links):
            raise ValueError("Duplicate link names")                  # urdf.py:1649-1652
        if len({j.name for j in self.joints}) != len(self.joints):
            raise ValueError("Duplicate joint names")
        for j in self.joints:
            child, parent = self.linkdict[j.child], self.linkdict[j.parent]
            child.parent, child.joint = parent, j
        for l in self.links:                 # siblings in LINK file order (BaseRobot._sort_links robot/BaseRobot.py:264-267 walks the link list)
            if l.parent is not None:
                l.parent.children.append(l)
        roots = [l for l in self.links if l.parent is None]
        if len(roots) != 1:
            raise ValueError("URDF must have exactly one root link, found %d" % len(roots))
        self.base_link = roots[0]
        # robot-wide joint numbering: the reference's -- depth first from the base link, children in the order their LINKS appear in the file
        # (BaseRobot._sort_links robot/BaseRobot.py:264-267, 336-350 on the link list urdf.py:1666-1676 builds in link order; the joints' order
        # does not enter) -- so that every method of this object, the
        # ERobot made from it (erobot(): the dynamics) and a reference Robot read the same column for the same joint.  (Until round 4 this was
        # the joints' FILE order: the same thing for a serial arm, not for a branched robot whose file lists one branch's tail after another
        # branch -- YuMi's grippers.)
        self.jindex = {}
        stack = [self.base_link]
        while stack:
            l = stack.pop()
            j = l.joint
            if j is not None and j.actuated and j.variable() is not None:
                self.jindex[j.name] = len(self.jindex)
            stack.extend(reversed(l.children))
        self.n = len(self.jindex)
        self.q = np.zeros(self.n)             # the stored configuration (BaseRobot.q): what q = None means (jacobm, gravload)
        self._cache = {}
        self.tool = tool
        self.base = None
        self._ee = ee

    # the gripper tool sits inside ets(None) (see ets); the base is applied by fkine alone, as in the reference (RobotKinematics.fkine hands it
    # to ETS.fkine, the Jacobians are in the robot's base frame).  Assigning either drops the kept chains.
    @property
    def tool(self): return self._tool_T
    @tool.setter
    def tool(self, T):
        if T is not None and hasattr(T, "A") and not isinstance(T, np.ndarray):
            T = T.A
        self._tool_T = None if T is None else _poses(np.asarray(T, dtype=np.float64).reshape(4, 4).copy())
        self._cache.clear()

    @property
    def base(self): return self._base
    @base.setter
    def base(self, T):
        if T is not None and hasattr(T, "A") and not isinstance(T, np.ndarray):
            T = T.A
        self._base = None if T is None else _poses(np.asarray(T, dtype=np.float64).reshape(4, 4).copy())

    # ------------------------------------------------------------ structure
    def path(self, end):
        """Links from the root (exclusive) to `end` (inclusive)."""
        link = self.linkdict[end] if isinstance(end, str) else end
        out = []
        while link.parent is not None:
            out.append(link)
            link = link.parent
        return out[::-1]

    def njoints(self, end):
        return sum(1 for l in self.path(end) if l.joint.actuated and l.joint.variable() is not None)

    @property
    def leaves(self):
        return [l for l in self.links if not l.children]

    @property
    def ee(self):
        """Default end link: the one given at construction, else the leaf whose path carries the most
        joints (first in URDF order on ties)."""
        if self._ee is not None:
            return self._ee
        best = max(self.leaves, key=lambda l: self.njoints(l))
        return best.name

    def ets(self, start=None, end=None, compact=True):
        """ETS from the root (or `start`) to `end` -- `ets(start, end)`, the reference's order (BaseRobot.ets robot/BaseRobot.py:1555);
        links as names or link objects.  compact=True numbers the joints 0..n-1 along the
        path (q is (N, n)); compact=False keeps the robot-wide jindex (q is (N, robot.n)).
        With no `end` the chain runs to the model's end effector AND carries the gripper's tool transform as its last element, as
        the reference's BaseRobot.ets does for a robot with one gripper (robot/BaseRobot.py:1610-1616, 1469-1473)."""
        with_tool = end is None and self.tool is not None
        end = self.ee if end is None else (end if isinstance(end, str) else end.name)
        start = start if (start is None or isinstance(start, str)) else start.name
        key = (end, start, compact, with_tool)
        if key in self._cache:
            return self._cache[key]
        links = self.path(end)
        if start is not None:
            names = [l.name for l in links]
            if start != self.base_link.name:
                if start not in names:
                    raise ValueError("start link %r is not an ancestor of %r" % (start, end))
                links = links[names.index(start) + 1:]
        ets, k = [], 0
        for l in links:
            ets.append(ET.SE3(l.joint.constant()))
            j = l.joint
            if j.actuated:
                var = j.variable(jindex=k if compact else self.jindex[j.name])
                if var is not None:
                    ets.append(var)
                    k += 1
        if with_tool:
            ets.append(ET.SE3(self.tool))
        e = ETS(ets)
        if not compact:
            e.q_width = self.n          # every branch reads the same (N, robot.n) array
        self._cache[key] = e
        return e

    def _tool_for(self, end, tool):
        """the tool a pass-through hands its chain: the caller's, nothing else.  The gripper's tool is part of ets(None) (the chain to the
        default end effector, robot/BaseRobot.py:1507-1521, 1610-1616) and of no other chain: with an explicit `end` link the reference
        applies none, and the IK entry points solve the same chains, so fkine and ik_* agree for every (start, end)."""
        return tool

    def _chain(self, q, start, end):
        """The chain a pass-through evaluates: joints numbered along the path (q has the path's columns) -- or, when q carries one column per
        joint of the WHOLE robot (`np.zeros(robot.n)`, as callers of the reference write: its chains address q by the robot-wide jindex),
        the same path with the robot-wide numbers."""
        e = self.ets(start, end)
        width = None if q is None else (q.shape[-1] if hasattr(q, "shape") and len(q.shape) else len(q))
        if width is not None and width == self.n and width != e.q_width:
            return self.ets(start, end, compact=False)
        return e

    def qlim(self, end=None):
        return self.ets(end=end).qlim

    # ------------------------------------------------------------ kinematics pass-throughs
    def fkine(self, q, end=None, start=None, tool=None, include_base=True):
        return self._chain(q, start, end).fkine(q, base=self.base, tool=self._tool_for(end, tool), include_base=include_base)

    def jacob0(self, q, end=None, start=None, tool=None):
        return self._chain(q, start, end).jacob0(q, tool=self._tool_for(end, tool))

    def jacobe(self, q, end=None, start=None, tool=None):
        return self._chain(q, start, end).jacobe(q, tool=self._tool_for(end, tool))

    def fkine_jacob0(self, q, end=None, start=None, tool=None):
        return self._chain(q, start, end).fkine_jacob0(q, tool=self._tool_for(end, tool))

    def ik_LM(self, Tep, end=None, start=None, **kw):
        return self.ets(start, end).ik_LM(Tep, **kw)

    # the other solvers and differential-kinematics methods of RobotKinematicsMixin (robot/RobotKinematics.py): self.ets(end).<name>(...)
    def ik_GN(self, Tep, end=None, start=None, **kw): return self.ets(start, end).ik_GN(Tep, **kw)
    def ik_NR(self, Tep, end=None, start=None, **kw): return self.ets(start, end).ik_NR(Tep, **kw)
    def ikine_LM(self, Tep, end=None, start=None, **kw): return self.ets(start, end).ikine_LM(Tep, **kw)
    def ikine_NR(self, Tep, end=None, start=None, **kw): return self.ets(start, end).ikine_NR(Tep, **kw)
    def ikine_GN(self, Tep, end=None, start=None, **kw): return self.ets(start, end).ikine_GN(Tep, **kw)
    def ikine_QP(self, Tep, end=None, start=None, **kw): return self.ets(start, end).ikine_QP(Tep, **kw)
    def hessian0(self, q=None, end=None, start=None, J0=None, tool=None): return self._chain(q, start, end).hessian0(q, J0=J0, tool=self._tool_for(end, tool))
    def hessiane(self, q=None, end=None, start=None, Je=None, tool=None): return self._chain(q, start, end).hessiane(q, Je=Je, tool=self._tool_for(end, tool))
    def manipulability(self, q=None, J=None, end=None, start=None, **kw):
        if J is not None:
            from .et import manipulability_from_jacobian
            return manipulability_from_jacobian(J, **kw)                # a pure function of J (robot/Robot.py:896)
        e = self.ets(start, end)                                     # Robot.manipulability: self.ets(end, start), gripper tool included (robot/Robot.py:825)
        return e.manipulability(np.zeros(e.n) if q is None else q, **kw)      # q=None: the robot's stored configuration, zeros (BaseRobot.q)

    # Robot.jacobm resolves `end` to a LINK first (robot/Robot.py:1182, _get_limit_links robot/BaseRobot.py:1478-1540) and drops the gripper
    # tool it returns, so its Jacobian is that of the end LINK's frame -- unlike robot.manipulability(q) and robot.ets().jacobm(q), whose
    # chain carries the tool.  (Only the translational measure can tell the two apart.)  Reproduced.

    def jacobm(self, q=None, J=None, H=None, end=None, start=None, **kw):
        e = self.ets(start, self.ee if end is None else end)
        if J is not None or H is not None:
            from .et import jacobm_from_jacobian
            if J is None:
                J = e.jacob0(self._stored_q(e) if q is None else q)          # robot/Robot.py:1194-1199
            return jacobm_from_jacobian(J, H=H, **kw)
        return e.jacobm(self._stored_q(e) if q is None else q, **kw)

    def _stored_q(self, e):
        """q = None means the robot's stored configuration `self.q` (BaseRobot.q, zeros until the user sets it; robot/Robot.py:1195-1198),
        read on the columns the path's joints use."""
        q = np.asarray(getattr(self, "q", None) if getattr(self, "q", None) is not None else np.zeros(e.n), dtype=np.float64).reshape(-1)
        if q.size == e.n:
            return q.copy()
        cols = [int(j) for j in e.jindices]
        return q[cols].copy() if q.size > max(cols, default=-1) else np.zeros(e.n)
    def jacob0_dot(self, q, qd, J0=None, representation=None, end=None): return self.ets(end=end).jacob0_dot(q, qd, J0=J0, representation=representation)
    def jacob0_analytical(self, q, representation="rpy/xyz", end=None, start=None, tool=None):
        return self.ets(start, end).jacob0_analytical(q, representation=representation, tool=self._tool_for(end, tool))
    def partial_fkine0(self, q, n=3, end=None, start=None): return self.ets(start, end).partial_fkine0(q, n)

    # ------------------------------------------------------------ dynamics
    def erobot(self, exclude=()):
        """The link tree as an rtbhip.ERobot (what URDF.__init__ + Robot.__init__ build in the reference:
        one Link per URDF link with ets = [SE3(constant) * joint ET], m, r from <inertial>).  `exclude`
        names links whose whole subtree is left out -- the reference removes gripper links from
        robot.links (BaseRobot.py:277-289), so Robot.rne never sees them."""
        from .erobot import ERobot
        return ERobot(self.erobot_links(exclude), name=self.name)

    def erobot_links(self, exclude=()):
        """The Link list erobot() constructs its robot from (Robot.URDF hands it to the caller's own class: robot/Robot.py:325-331)."""
        from .erobot import Link
        drop = set()

        def mark(l):
            drop.add(l.name)
            for c in l.children:
                mark(c)
        for nm in exclude:
            mark(self.linkdict[nm])
        made = {}
        links = []
        for l in self.links:                                  # URDF file order (parents may come later in the file)
            if l.name in drop:
                continue
            ets = []
            if l.joint is not None:
                ets.append(ET.SE3(l.joint.constant()))
                var = l.joint.variable()
                if var is not None:
                    ets.append(var)
            k = Link(ets=ETS(ets), m=l.m, r=l.r, I=l.I, name=l.name)
            made[l.name] = k
            links.append(k)
        for l in self.links:
            if l.name in made and l.parent is not None:
                made[l.name].parent = made[l.parent.name]
        return links

    def fkine_all(self, q, exclude=()):
        """Pose of every link frame (reference Robot.fkine_all robot/Robot.py:638-698): see ERobot.fkine_all."""
        return self._tree(exclude).fkine_all(q)

    def _tree(self, exclude=()):
        key = ("erobot", tuple(exclude))
        if key not in self._cache:
            self._cache[key] = self.erobot(exclude)
        return self._cache[key]

    def rne(self, q, qd=None, qdd=None, symbolic=False, gravity=None, exclude=()):
        return self._tree(exclude).rne(q, qd, qdd, symbolic=symbolic, gravity=gravity)           # (Robot.rne's order, robot/Robot.py:1704)

    # the Dynamics-mixin terms (reference robot/Dynamics.py over Robot.rne): see ERobot
    def inertia(self, q, exclude=()):
        return self._tree(exclude).inertia(q)

    def coriolis(self, q, qd, exclude=()):
        return self._tree(exclude).coriolis(q, qd)

    def gravload(self, q=None, gravity=None, exclude=()):
        t = self._tree(exclude)
        return t.gravload(np.zeros(t.n) if q is None else q, gravity=gravity)      # q=None: the stored configuration, zeros (BaseRobot.q)

    def itorque(self, q, qdd, exclude=()):
        return self._tree(exclude).itorque(q, qdd)

    def accel(self, q, qd, torque, gravity=None, exclude=()):
        return self._tree(exclude).accel(q, qd, torque, gravity=gravity)


def loadstr(urdf_string, **kw):
    return URDFRobot(urdf_string, **kw)


# The robot descriptions the reference names relative to its data package (rtb-data/rtbdata/xacro/<path>, what models/URDF/<name>.py hand to
# Robot.URDF_read) -> the kinematic URDF of the same robot shipped here (expanded offline by scripts/make_urdf_data.py with the reference's own
# xacro tool): the data package itself is not part of this backend, so these names work without it.  A file that IS on disk is read at run
# time, xacro included (rtbhip.xacro).
REFERENCE_PATHS = {
    "al5d_description/urdf/al5d_robot.urdf": "AL5D", "fetch_description/robots/fetch.urdf": "Fetch",
    "kortex_description/robots/gen3.xacro": "KinovaGen3", "kuka_description/kuka_lbr_iiwa/urdf/lbr_iiwa_14_r820.xacro": "LBR",
    "kinova_description/urdf/j2n4s300_standalone.xacro": "Mico", "franka_description/robots/panda_arm_hand.urdf.xacro": "Panda",
    "puma560_description/urdf/puma560_robot.urdf.xacro": "Puma560", "ur_description/urdf/ur3_joint_limited_robot.urdf.xacro": "UR3",
    "ur_description/urdf/ur5_joint_limited_robot.urdf.xacro": "UR5", "ur_description/urdf/ur10_joint_limited_robot.urdf.xacro": "UR10",
    "yumi_description/urdf/yumi.urdf": "YuMi",
    **{"interbotix_descriptions/urdf/%s.urdf.xacro" % k: k for k in ("px100", "px150", "rx150", "rx200", "vx300", "vx300s", "wx200", "wx250", "wx250s")},
}


def read(file_path, mappings=None, packages=None, **kw):
    """The robot description at `file_path` as a URDFRobot (the reader behind Robot.URDF, robot/Robot.py:218-330).  A path that exists is read:
    plain URDF as it is, a `.xacro` file through rtbhip.xacro (`mappings`: values for its xacro:arg's, `packages`: where `$(find pkg)` looks;
    robot/Robot.py:262-275 does the same with the reference's bundled tool).  A path the reference resolves inside its data package
    ("fetch_description/robots/fetch.urdf", REFERENCE_PATHS) that is not on disk gives the shipped description of that robot."""
    path = os.fspath(file_path)
    if os.path.isfile(path):
        if path.endswith(".xacro"):
            from . import xacro
            return URDFRobot(xacro.process(path, mappings=mappings, packages=packages), **kw)
        with open(path) as f:
            return URDFRobot(f.read(), **kw)
    key = path.replace(os.sep, "/").lstrip("./")
    if key in REFERENCE_PATHS:
        shipped = os.path.join(DATA_DIR, REFERENCE_PATHS[key] + ".urdf")
        with open(shipped) as f:
            return URDFRobot(f.read(), **kw)
    raise FileNotFoundError("no URDF file %r (and not one of the reference's data-package paths this backend ships: %s)"
                            % (path, ", ".join(sorted(REFERENCE_PATHS))))


def available():
    return sorted(f[:-5] for f in os.listdir(DATA_DIR) if f.endswith(".urdf"))


# end links / gripper tools of the reference's model classes (models/URDF/*.py `gripper_links`)
_MODEL_EE = {
    "Panda": ("panda_hand", [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0.1034], [0, 0, 0, 1]]),   # models/URDF/Panda.py:43-48
}


# named joint configurations of the reference's model classes (models/URDF/Panda.py:54-55, UR5.py:48-68, Puma560.py:57-91): vectors over the
# joints of the arm (the path to the model's end effector), set as attributes by load()
_pi = np.pi
_MODEL_Q = {
    "Panda": dict(qr=[0, -0.3, 0, -2.2, 0, 2.0, _pi / 4], qz=[0.0] * 7),
    "UR5": dict(qr=[_pi, 0, 0, 0, _pi / 2, 0], qz=[0.0] * 6, q1=[0, -_pi / 2, _pi / 2, 0, _pi / 2, 0]),
    "Puma560": dict(qr=[0, _pi / 2, -_pi / 2, 0, 0, 0], qz=[0.0] * 6, ru=[-0.0, 0.7854, 3.1416, -0.0, 0.7854, 0.0],
                    rd=[-0.0, -0.8335, 0.0940, -3.1416, 0.8312, 3.1416], lu=[2.6486, -3.9270, 0.0940, 2.5326, 0.9743, 0.3734],
                    ld=[2.6486, -2.3081, 3.1416, 0.6743, 0.8604, 2.6611], qs=[0, 0, -_pi / 2, 0, 0, 0], qn=[0, _pi / 4, _pi, 0, _pi / 4, 0]),
}


def load(name, **kw):
    """One of the pre-expanded robot descriptions in rtbhip/data/urdf (see available())."""
    path = os.path.join(DATA_DIR, name + ".urdf")
    if not os.path.exists(path):
        raise ValueError("unknown robot %r; available: %s" % (name, ", ".join(available())))
    if name in _MODEL_EE and "ee" not in kw:
        kw["ee"], kw["tool"] = _MODEL_EE[name][0], np.array(_MODEL_EE[name][1], dtype=np.float64)
    robot = URDFRobot(open(path).read(), name=name, **kw)
    for key, v in _MODEL_Q.get(name, {}).items():
        setattr(robot, key, np.array(v, dtype=np.float64))
    return robot


# the 16 arms of BASELINE config 5 ("mixed fleet", 4..10 joints on the path to the deepest leaf)
FLEET16 = ["AL5D", "px100", "px150", "rx200", "vx300s", "wx250s", "UR3", "UR5", "UR10", "Puma560", "LBR", "Panda",
           "KinovaGen3", "Mico", "Fetch", "YuMi"]

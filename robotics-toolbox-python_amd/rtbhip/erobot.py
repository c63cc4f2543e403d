"""Link / ERobot -- ETS robots (link trees) with batched inverse dynamics on the GPU.

Mirrors the part of the reference's ``Link`` (robot/Link.py) and ``ERobot`` / ``Robot`` (robot/ERobot.py,
robot/Robot.py, robot/BaseRobot.py) that ``Robot.rne`` (robot/Robot.py:1704-1903) touches: links with an
ETS whose last element may be a joint, mass ``m``, centre of mass ``r``, a parent link; link ordering and
automatic joint numbering by depth-first traversal (BaseRobot._sort_links, robot/BaseRobot.py:198-340);
the grouping of static links with the next joint (Robot.py:1777-1789).  The dynamics run in
``k_tree_rne`` (csrc/tree_kernels.hip) behind ``rtbhip_tree_rne``; nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, as_numeric, host_ptr, is_torch, MEM_HOST, MEM_DEVICE
from .et import ET, ETS, _AXES
from .kinematics import RobotKinematics, as_se3
from .linkdyn import LinkDynamics


class Link(LinkDynamics):
    """reference robot/Link.py: Link(ets=ETS(...), m=, r=, I=, Jm=, G=, B=, Tc=, parent=, name=, jindex=, qlim=)"""

    def __init__(self, ets=None, jindex=None, name=None, parent=None, joint_name=None, m=None, r=None, I=None, Jm=None, B=None, Tc=None, G=None,
                 qlim=None, **kw):
        # (positional order of the reference: Link(ets, jindex, **kwargs) over BaseLink(ets, name, parent, joint_name, m, r, I, Jm, B, Tc, G,
        # qlim, ...), robot/Link.py:1556-1580, :70-110)
        if ets is None:
            ets = ETS()
        elif isinstance(ets, ET):
            ets = ETS(ets)
        elif not isinstance(ets, ETS):
            raise TypeError("The ets argument must be of type ETS or ET")                  # robot/Link.py:189-196
        if parent is not None and not isinstance(parent, (str, Link)):
            raise TypeError("parent must be BaseLink subclass")                            # robot/Link.py:200-210
        self.ets = ets
        self._set_dynamics(m=m, r=r, I=I, Jm=Jm, G=G, B=B, Tc=Tc)
        self.parent = parent
        self.name = name if name is not None else ""
        self._joint_name = joint_name
        self.jindex = jindex if jindex is not None else (ets[-1].jindex if self.isjoint else None)
        self._qlim = None
        if qlim is not None and self.isjoint:
            self.qlim = qlim
        self.children = []
        self.robot = None
        self.number = 0

    _DYN = ("m", "r", "I", "Jm", "G", "B", "Tc")

    def __setattr__(self, name, value):
        """A changed dynamic parameter invalidates the owning robot's device link table (the reference's `_listen_dyn`, robot/Link.py:28-45)."""
        object.__setattr__(self, name, value)
        if name in Link._DYN:
            robot = self.__dict__.get("robot")
            if robot is not None and hasattr(robot, "dynchanged"):
                robot.dynchanged()
        elif name in ("_ets", "_qlim", "jindex", "parent"):
            # geometry: the chains the owning robot keeps per (start, end), their device tables and its tree table were made from the old value
            robot = self.__dict__.get("robot")
            if robot is not None and hasattr(robot, "_kinchanged"):
                robot._kinchanged()

    @property
    def ets(self): return self._ets
    @ets.setter
    def ets(self, ets):
        joints = [k for k, e in enumerate(ets) if e.isjoint]
        if len(joints) > 1:
            raise ValueError("An elementary link can only have one joint variable")       # robot/Link.py:230-240
        if joints and joints[0] != len(ets) - 1:
            raise ValueError("Variable link must be at the end of the ETS")
        self._ets = ets
        self.isjoint = bool(joints)

    @property
    def v(self):
        return self.ets[-1] if self.isjoint else None

    @property
    def qlim(self):
        """The joint's limits, kept on the link (robot/Link.py:1010-1040 keeps them on the joint's ET)."""
        if self._qlim is not None:
            return self._qlim
        return self.v.qlim if self.isjoint else None
    @qlim.setter
    def qlim(self, v):
        if v is None:
            self._qlim = None
            return
        if not self.isjoint:
            raise ValueError("Can not set qlim on a static joint")
        self._qlim = np.asarray(v, dtype=np.float64).reshape(2).copy()

    @property
    def isrevolute(self): return self.isjoint and self.v.isrotation
    @property
    def isprismatic(self): return self.isjoint and self.v.istranslation
    @property
    def isflip(self): return self.isjoint and self.v.isflip
    @property
    def nchildren(self): return len(self.children)

    def copy(self, parent=None):
        import copy as _copy
        robot, par, kids = self.robot, self.parent, self.children
        self.robot, self.parent, self.children = None, None, []
        try:
            l = _copy.deepcopy(self)
        finally:
            self.robot, self.parent, self.children = robot, par, kids
        l.parent = parent
        return l

    def Ts(self):
        """Constant part of the link transform (robot/Link.py:1642-1651), 4x4."""
        T = np.eye(4)
        for e in self.ets:
            if not e.isjoint:
                T = T @ e.T
        return T

    def A(self, q=0.0):
        """The link transform at joint value q (robot/Link.py:1590-1640: Ts, times the joint's transform when there is one), evaluated
        on the device as the link's own chain."""
        ets = self.ets
        if self.isjoint and self.v.jindex not in (None, 0):                # the chain of this link alone reads a one-column q
            ets = ETS([ET(e.axis, flip=e.isflip, jindex=0, qlim=e.qlim) if e.isjoint else e for e in ets])
        return ets.fkine(np.array([float(q)]) if self.isjoint else np.zeros(0))

    def __str__(self):
        """`Link("name", Rx(88.41°) ⊕ tz(1))` (robot/Link.py:335-369)"""
        s = type(self).__name__ + "("
        if self.name is not None:
            s += '"%s"' % self.name
        if len(self.ets) > 0:
            s += ", %s" % self.ets
        return s + ")"


class ERobot(RobotKinematics):
    """ERobot(links) or ERobot(ets) (reference Robot.__init__ robot/Robot.py:60-160): a list of Link objects, or an ETS that is cut
    into one link per joint ("a link frame after every joint", named link0, link1, ...; :116-131).  Kinematics: the
    RobotKinematics surface over ets(start, end); dynamics: rne."""

    def __init__(self, links, gripper_links=None, name="", manufacturer="", comment="", base=None, tool=None, gravity=(0, 0, -9.81), keywords=(),
                 symbolic=False, configs=None, check_jindex=True, urdf_string=None, urdf_filepath=None, **kw):
        # (the reference's positional order, robot/BaseRobot.py:95-112 / robot/Robot.py:60-80)
        self.comment, self.keywords = comment, tuple(keywords)
        if kw:
            raise TypeError("unexpected keyword argument(s): %s" % ", ".join(sorted(kw)))
        if gripper_links is not None:
            raise TypeError("unexpected keyword argument(s): gripper_links")      # grippers are outside this backend
        if symbolic:
            raise TypeError("Symbolic value")                                      # symbolic robots stay on the reference's Python path
        self.urdf_string, self.urdf_filepath = urdf_string, urdf_filepath
        self.configs = dict(configs) if configs else {}
        if isinstance(links, ET):
            links = ETS(links)
        if isinstance(links, ETS):
            parent, cut = None, []
            for j, seg in enumerate(links.split()):
                parent = Link(seg, parent=parent, name="link%d" % j)
                cut.append(parent)
            links = cut
        links = list(links)
        names = {}
        for k, l in enumerate(links):
            if not isinstance(l, Link):
                raise TypeError("links should all be Link subclass")
            if not l.name:
                l.name = "link-%d" % k
            if l.name in names:
                raise ValueError("link name %s is not unique" % l.name)
            names[l.name] = l
            l.children = []
        for l in links:
            if isinstance(l.parent, str):
                l.parent = names[l.parent]
        if all(l.parent is None for l in links):                    # BaseRobot.py:243-245: a serial chain in list order
            for a, b in zip(links[:-1], links[1:]):
                b.parent = a
        bases = [l for l in links if l.parent is None]
        if len(bases) != 1:
            raise ValueError("Multiple base links" if bases else "Invalid link configuration provided, must have a base link")
        for l in links:
            if l.parent is not None:
                l.parent.children.append(l)
        # depth-first order and automatic joint numbering (BaseRobot.py:316-330, dfs_links :1842)
        order = []
        stack = [bases[0]]
        while stack:
            l = stack.pop()
            order.append(l)
            stack.extend(reversed(l.children))
        auto = all(l.jindex is None for l in order if l.isjoint)
        if auto:
            k = 0
            for l in order:
                if l.isjoint:
                    l.jindex = k
                    k += 1
        elif any(l.jindex is None for l in order if l.isjoint):
            raise ValueError("all links must have a jindex, or none have a jindex")
        else:
            order = links                                            # explicit numbering keeps the given order (:355)
        self.links = order
        for k, l in enumerate(links):
            l.robot, l.number = self, k + 1                            # BaseRobot.py:234-236: numbered in the order GIVEN, before the sort
        self.name = name
        self.manufacturer = manufacturer
        self.base = as_se3(base, "base")
        self.tool = as_se3(tool, "tool")
        self.base_link = bases[0]
        self.n = sum(1 for l in order if l.isjoint)
        if sorted(l.jindex for l in order if l.isjoint) != list(range(self.n)):
            raise ValueError("joint index was repeated or out of range")
        self.gravity = np.asarray(gravity, dtype=np.float64).reshape(3)
        self._tree = None

    @classmethod
    def URDF(cls, file_path, gripper=None):
        """Robot.URDF(file_path, gripper=None) (robot/Robot.py:288-330): the robot a URDF file describes, one Link per URDF link in file
        order.  `gripper` (an index into that list, or a link name): that link and everything beyond it are the gripper and leave the
        rigid-body tree (BaseRobot.py:288-314), so `n` counts the joints before it.  Plain URDF files, and the paths the reference resolves
        inside its data package (rtbhip.urdf.read)."""
        from . import urdf
        u = urdf.read(file_path)
        exclude = ()
        if gripper is not None:
            if isinstance(gripper, bool) or not isinstance(gripper, (int, str)):
                raise TypeError("bad argument passed as gripper")
            if isinstance(gripper, int):
                exclude = (u.links[gripper].name,)
            elif gripper in u.linkdict:
                exclude = (gripper,)
            else:
                raise ValueError("no link named %s" % gripper)
        # constructed THROUGH the caller's class, as the reference does (`return cls(links, name=..., urdf_string=..., urdf_filepath=...)`,
        # robot/Robot.py:325-331): a subclass's __init__ runs.  (The gripper's links are already out of the list: this backend has no Gripper objects.)
        return cls(u.erobot_links(exclude), name=u.name, urdf_string=u.urdf_string, urdf_filepath=file_path)

    def __len__(self): return len(self.links)
    def __getitem__(self, i): return self.links[i]

    @property
    def gravity(self): return self._gravity
    @gravity.setter
    def gravity(self, g):
        self._gravity = np.asarray(g, dtype=np.float64).reshape(3).copy()          # robot/BaseRobot.py:903-906

    # ------------------------------------------------------------ kinematics over a path
    def _getlink(self, link, default):
        if link is None:
            return default
        if isinstance(link, Link):
            if link not in self.links:
                raise ValueError("link not in robot links")              # BaseRobot.py:1407-1410
            return link
        if isinstance(link, str):
            for l in self.links:
                if l.name == link:
                    return l
            raise ValueError("no link named %s" % link)
        raise TypeError("unknown argument")

    def _link_ets(self, l):
        """A link's ETS with the robot-wide joint number on its joint."""
        return [ET(e.axis, flip=e.isflip, jindex=l.jindex, qlim=l.qlim) if e.isjoint else e for e in l.ets]     # l.qlim: the link-level limit first (robot/Link.py:1010-1040 writes through to the joint ET)

    def ets(self, start=None, end=None):
        """ETS of the path from link `start` (default: the base link) to link `end` (default: the last link), links given as
        Link objects or names; robot-wide jindex kept.  The reference's rule (BaseRobot.ets robot/BaseRobot.py:1555-1652 ->
        _find_ets :1426-1467): the path INCLUDES the start link's own transform when it descends from it, and climbing
        towards the root multiplies by the inverse of each link left behind."""
        a = self._getlink(start, self.base_link)
        b = self._getlink(end, self.links[-1])
        made = self.__dict__.setdefault("_ets_made", {})
        if (id(a), id(b)) in made:
            return made[(id(a), id(b))]            # one ETS object (and one device chain table) per path
        up, l = [], a
        anc_a = []
        while l is not None:
            anc_a.append(l)
            l = l.parent
        down, l = [], b
        while l is not None and l not in anc_a:
            down.append(l)
            l = l.parent
        if l is None:
            raise ValueError("Could not find the requested ETS in this robot")
        meet = l                                                   # lowest common ancestor (a itself when b is below a)
        out = []
        if meet is a:
            out += self._link_ets(a)                               # toplevel: path = link.ets (:1445-1446)
        else:
            l = a
            while l is not meet:                                   # climbing: link.ets.inv() of every link left (:1459-1465)
                out += [e.inv() for e in reversed(self._link_ets(l))]
                l = l.parent
        for l in reversed(down):
            out += self._link_ets(l)
        made[(id(a), id(b))] = ETS(out)
        if made[(id(a), id(b))].n and self.n <= 256:
            made[(id(a), id(b))].q_width = self.n            # every path reads the robot's q rows as they are (no column copy per call)
        return made[(id(a), id(b))]

    @property
    def qlim(self):
        """(2, n) joint limits, one column per joint link IN LINK ORDER (reference BaseRobot.qlim robot/BaseRobot.py:979-1038 fills column j for the
        j-th joint link it meets -- the joint numbers for every automatically numbered robot, not for one numbered by hand); an unset or NaN
        revolute limit reads as [-pi, pi], an unset prismatic one raises."""
        lim = np.zeros((2, self.n))
        j = 0
        for l in self.links:
            if not l.isjoint:
                continue
            ql = l.qlim
            if l.isrevolute:
                if ql is None or np.any(np.isnan(np.asarray(ql, dtype=np.float64))):
                    ql = (-np.pi, np.pi)
            elif ql is None:
                raise ValueError("Undefined prismatic joint limit")
            lim[:, j] = np.asarray(ql, dtype=np.float64).reshape(2)
            j += 1
        return lim

    @qlim.setter
    def qlim(self, new_qlim):
        new_qlim = np.array(new_qlim)
        if new_qlim.shape != (2, self.n):
            raise ValueError("new_qlim must be of shape (2, n)")
        j = 0
        for l in self.links:                       # robot/BaseRobot.py:1041-1051: column j to the j-th joint link
            if l.isjoint:
                l.qlim = new_qlim[:, j]
                j += 1

    def fkine_all(self, q, base=None):
        """Pose of every link frame: T[0] = base, T[link.number] = that link -- `number` is the link's position in the list the robot
        was built from, plus one (robot/Robot.py:679 indexes by it; for a serial chain that is its place in self.links);
        (nlinks+1,4,4) or (N,nlinks+1,4,4)
        (reference Robot.fkine_all robot/Robot.py:638-698, a Python recursion over the link tree).  One chain walk per
        leaf of the tree: the frames of all the links on its path come out of the same launch."""
        index = {id(l): k for k, l in enumerate(self.links)}
        done = {}
        is_torch = type(q).__module__.startswith("torch")
        for leaf in (l for l in self.links if not l.children):
            path = []
            l = leaf
            while l is not None:
                path.append(l)
                l = l.parent
            path.reverse()
            marks, want, k = [], [], 0
            for l in path:
                k += len(l.ets)
                if id(l) not in done:
                    marks.append(k); want.append(l)
            if not marks:
                continue
            e = self.ets(end=leaf)
            qa = q if is_torch else np.asarray(q, dtype=np.float64)
            F = e.link_frames(qa[..., :e.q_width], marks, base=base)       # the path reads the robot-wide joint columns it uses
            for m, l in enumerate(want):
                done[id(l)] = F[..., m, :, :]
        first = next(iter(done.values()))
        if is_torch:
            import torch
            T0 = torch.eye(4, dtype=first.dtype, device=first.device) if base is None else torch.as_tensor(np.asarray(base, dtype=np.float64)).to(first.device)
            T0 = T0.expand(first.shape)
            return torch.stack([T0] + [done[id(l)] for l in sorted(self.links, key=lambda l: l.number)], dim=-3)
        T0 = np.broadcast_to(np.eye(4) if base is None else np.asarray(base, dtype=np.float64), first.shape)
        return np.stack([T0] + [done[id(l)] for l in sorted(self.links, key=lambda l: l.number)], axis=-3)

    # ------------------------------------------------------------ dynamics
    def link_groups(self):
        """Robot.py:1777-1789: static links are grouped with the first joint that follows them in link order."""
        groups, cur = [], []
        for i, l in enumerate(self.links):
            cur.append(i)
            if l.isjoint:
                groups.append(cur)
                cur = []
        return groups

    def group_table(self):
        """One record per link group for rtbhip_tree_create (see include/rtbhip.h)."""
        groups = self.link_groups()
        where = {}
        for g, idx in enumerate(groups):
            for i in idx:
                where[i] = g
        index_of = {id(l): i for i, l in enumerate(self.links)}
        recs = []
        for g, idx in enumerate(groups):
            T = np.eye(4)
            m, h, I = 0.0, np.zeros(3), np.zeros((3, 3))
            for i in idx:
                l = self.links[i]
                T = T @ l.Ts()
                m += l.m                                             # SpatialInertia(m, r) summed as is (Robot.py:1793-1800)
                h += l.m * l.r
                I += l.m * (np.dot(l.r, l.r) * np.eye(3) - np.outer(l.r, l.r))
            first, joint = self.links[idx[0]], self.links[idx[-1]]
            parent = -1
            if first.parent is not None:
                pi = index_of[id(first.parent)]
                if pi not in where:
                    raise ValueError("parent link %s carries no joint up to the end of the link list" % first.parent.name)
                parent = where[pi]
            recs.append(dict(parent=parent, kind=_AXES[joint.v.axis], flip=int(joint.v.isflip), jindex=int(joint.jindex), T=T,
                             m=m, h=h, I=np.array([I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]])))
        return recs

    def _handle(self):
        if self._tree is None:
            recs = self.group_table()
            arr = (_lib.rtbhip_tree_group * len(recs))()
            for k, r in enumerate(recs):
                arr[k].parent, arr[k].kind, arr[k].flip, arr[k].jindex = r["parent"], r["kind"], r["flip"], r["jindex"]
                arr[k].T[:] = list(np.ascontiguousarray(r["T"]).reshape(16))
                arr[k].m = r["m"]
                arr[k].h[:] = list(r["h"])
                arr[k].I[:] = list(r["I"])
            h = C.c_uint64(0)
            check(lib().rtbhip_tree_create(arr, len(recs), C.byref(h)))
            self._tree = h.value
        return self._tree

    def upload(self, device=None):
        """The link-group table resident on `device` now (rtbhip_tree_upload)."""
        check(lib().rtbhip_tree_upload(self._handle(), -1 if device is None else int(getattr(device, "index", device) or 0)))
        return self

    def dynchanged(self, what=None):
        if self._tree is not None and _lib._lib is not None:
            _lib._lib.rtbhip_tree_destroy(self._tree)
        self._tree = None

    def _kinchanged(self):
        """A link's geometry (ets, qlim, jindex, parent) was reassigned: drop the chains kept per path, their device tables and the tree table
        (DHRobot does the same for its links; the reference rebuilds `ets()` on every call)."""
        self.__dict__.pop("_ets_made", None)
        if hasattr(self, "_paths_changed"):
            self._paths_changed()
        if "_tree" in self.__dict__:
            self.dynchanged()

    def __getstate__(self):
        """copy.copy / copy.deepcopy / pickle: the device link-group table and the kept path chains belong to THIS robot; a copy builds its own."""
        state = dict(self.__dict__)
        state["_tree"] = None
        state.pop("_ets_made", None)
        state.pop("_path_cache", None)
        return state

    def __del__(self):
        try:
            self.dynchanged()
        except Exception:
            pass

    def rne(self, q, qd=None, qdd=None, symbolic=False, gravity=None):
        """Inverse dynamics: (n,) or (N,n) (reference Robot.rne robot/Robot.py:1704-1903)."""
        if symbolic:
            raise TypeError("Symbolic value")          # symbolic dynamics stay on the reference's Python path
        n = self.n
        first = q
        tm = is_torch(first) and first.is_cuda
        if tm:
            single = q.dim() == 1
            arrs = [None if x is None else x.reshape(-1, n).contiguous() for x in (q, qd, qdd)]
        else:
            single = as_numeric(q).ndim == 1
            arrs = [None if x is None else np.ascontiguousarray(as_numeric(x).reshape(-1, n)) for x in (q, qd, qdd)]
        N = arrs[0].shape[0]
        single = single or N == 1                      # Robot.py:1900-1903
        if any(x is not None and tuple(x.shape) != (N, n) for x in arrs):
            raise ValueError("q, qd, qdd must all be (%d,) or (N,%d)" % (n, n))
        g = np.ascontiguousarray(self.gravity if gravity is None else np.asarray(gravity, dtype=np.float64).reshape(3))
        if tm:
            import torch
            tau = torch.empty((N, n), dtype=torch.float64, device=arrs[0].device)
            ptr = lambda x: None if x is None else C.c_void_p(x.data_ptr())
            _lib.note_device(arrs[0])
            stream, mem = _lib.current_stream_ptr(), MEM_DEVICE
        else:
            tau = _lib.host_empty((N, n))
            ptr, stream, mem = host_ptr, None, MEM_HOST
        check(lib().rtbhip_tree_rne(self._handle(), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), N, host_ptr(g), ptr(tau), mem, stream))
        return tau[0] if single else tau

    # ---- the Dynamics-mixin terms (reference robot/Dynamics.py, which Robot inherits through BaseRobot): one fused kernel each
    def _dyn_args(self, arrays):
        """(arrays as (N,n), N, single, torch?, ptr, stream, mem, empty(shape))"""
        n = self.n
        tm = is_torch(arrays[0]) and arrays[0].is_cuda
        if tm:
            single = arrays[0].dim() == 1
            arrs = [x.reshape(-1, n).contiguous() for x in arrays]
        else:
            single = as_numeric(arrays[0]).ndim == 1
            arrs = [np.ascontiguousarray(as_numeric(x).reshape(-1, n)) for x in arrays]
        N = arrs[0].shape[0]
        if any(tuple(x.shape) != (N, n) for x in arrs):
            raise ValueError("arguments must all be (%d,) or (N,%d)" % (n, n))
        if tm:
            import torch
            _lib.note_device(arrs[0])
            dev = arrs[0].device
            return (arrs, N, single or N == 1, lambda x: C.c_void_p(x.data_ptr()), _lib.current_stream_ptr(), MEM_DEVICE,
                    lambda shape: torch.empty(shape, dtype=torch.float64, device=dev))
        return arrs, N, single or N == 1, host_ptr, None, MEM_HOST, _lib.host_empty

    def gravload(self, q=None, gravity=None):
        """tau_g(q) (reference Dynamics.gravload robot/Dynamics.py:863-922): rne(q, 0, 0)."""
        return self.rne(q, None, None, gravity=gravity)

    def itorque(self, q, qdd):
        """M(q) qdd (reference Dynamics.itorque robot/Dynamics.py:1407-1465): rne(q, 0, qdd) without gravity."""
        return self.rne(q, None, qdd, gravity=[0, 0, 0])

    def inertia(self, q):
        """Joint-space inertia matrix: (n,n) or (N,n,n) (reference Dynamics.inertia robot/Dynamics.py:704-763: n Robot.rne calls per
        configuration, row i = rne(q, 0, e_i) without gravity; here all passes of a configuration in one lane, csrc/tree_device.h)."""
        arrs, N, single, ptr, stream, mem, empty = self._dyn_args([q])
        M = empty((N, self.n, self.n))
        check(lib().rtbhip_tree_inertia(self._handle(), ptr(arrs[0]), N, ptr(M), mem, stream))
        return M[0] if single else M

    def coriolis(self, q, qd):
        """Coriolis / centripetal matrix C(q, qd): (n,n) or (N,n,n) (reference Dynamics.coriolis robot/Dynamics.py:765-861:
        n + n(n-1)/2 Robot.rne calls per configuration; here n two-field passes, column k = B(qd, e_k): csrc/tree_device.h tree_bilinear_core)."""
        arrs, N, single, ptr, stream, mem, empty = self._dyn_args([q, qd])
        Cm = empty((N, self.n, self.n))
        check(lib().rtbhip_tree_coriolis(self._handle(), ptr(arrs[0]), ptr(arrs[1]), N, ptr(Cm), mem, stream))
        return Cm[0] if single else Cm

    def accel(self, q, qd, torque, gravity=None):
        """Forward dynamics qdd = M(q)^-1 (torque - rne(q, qd, 0)): (n,) or (N,n) (reference Dynamics.accel robot/Dynamics.py:424-509)."""
        arrs, N, single, ptr, stream, mem, empty = self._dyn_args([q, qd, torque])
        g = np.ascontiguousarray(self.gravity if gravity is None else np.asarray(gravity, dtype=np.float64).reshape(3))
        qdd = empty((N, self.n))
        check(lib().rtbhip_tree_accel(self._handle(), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), N, host_ptr(g), ptr(qdd), mem, stream))
        return qdd[0] if single else qdd

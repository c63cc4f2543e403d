"""PoELink / PoERevolute / PoEPrismatic / PoERobot -- product-of-exponentials robots on the batched GPU path.

Mirrors reference robot/PoERobot.py: a robot is a list of joint twists given in the BASE frame plus the end-effector pose T0
at q = 0, T(q) = exp([S_1] q_1) ... exp([S_n] q_n) T0.  The reference evaluates fkine / jacob0 / jacobe of that product in Python,
one matrix exponential and one adjoint per joint per configuration (PoERobot.py:209-270); here the twists are lowered once, on
the host, to the canonical chain form of librtbhip (`rtbhip_chain_create_poe`, csrc/chain.cpp `compile_poe`: with W_i a frame
on the screw axis, exp([S_i] q) = W_i Z(q) W_i^-1, so the product telescopes into n z-joints between n + 1 constants) and every
call is a kernel launch over (N, n) configurations -- the same kernels that serve ETS and DH robots.

`PoERobot.ets()` returns what the reference's `ets()` returns -- the re-expression of the robot as elementary transforms that
`_update_ets` (PoERobot.py:272-324) builds through roll-pitch-yaw angles -- as an `rtbhip.ETS`; `Robot(r.ets())` of the
reference's test (tests/test_PoERobot.py:27) is `rtbhip.ERobot(r.ets())` here.

No arithmetic on q happens in this module: the host code below only prepares constants at construction time.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import check, lib, host_ptr
from .et import ET, ETS
from .kinematics import RobotKinematics, as_se3


def _unit(v, what):
    v = np.asarray(v, dtype=np.float64).reshape(-1)
    if v.shape != (3,):
        raise ValueError("%s must be a 3-vector" % what)
    nrm = float(np.linalg.norm(v))
    if not nrm > 0.0:
        raise ValueError("%s must be non-zero" % what)
    return v / nrm


def _se3_log(T):
    """The twist (v, w) with exp([S]) = T: what SE3.twist() hands PoELink for the end-effector link (PoERobot.py:175).  Only
    kept as the link's `S` attribute; the kinematics use T0 itself (PoERobot.py:286)."""
    R, t = T[:3, :3], T[:3, 3]
    c = min(1.0, max(-1.0, (np.trace(R) - 1.0) / 2.0))
    th = math.acos(c)
    if th < 1e-12:
        return np.r_[t, np.zeros(3)]
    if math.pi - th < 1e-9:                                   # half turn: the axis from the symmetric part
        k = int(np.argmax(np.diag(R)))
        a = (R[:, k] + np.eye(3)[k]) / math.sqrt(2.0 * (1.0 + R[k, k]))
        w = a * th
    else:
        w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) * (th / (2.0 * math.sin(th)))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Vinv = np.eye(3) - 0.5 * K + (1.0 / th ** 2) * (1.0 - th * math.sin(th) / (2.0 * (1.0 - math.cos(th)))) * (K @ K)
    return np.r_[Vinv @ t, w]


class PoELink:
    """A link defined by a twist S = (v, w) in the base frame (reference PoELink, PoERobot.py:10-121).  A plain PoELink is not a
    joint (the base and end-effector links of a PoERobot); PoERevolute / PoEPrismatic are."""
    isrevolute = False
    isprismatic = False

    def __init__(self, twist, name=None):
        S = np.asarray(getattr(twist, "S", twist), dtype=np.float64).reshape(-1)
        if S.shape != (6,):
            raise ValueError("a twist is a 6-vector (v, w)")
        self.S = S
        self.name = name

    @property
    def isjoint(self):
        return self.isrevolute or self.isprismatic

    @property
    def v(self): return self.S[:3]

    @property
    def w(self): return self.S[3:]

    def __repr__(self):
        s = "%s(%s" % (type(self).__name__, np.array2string(self.S, separator=","))
        if self.name is not None:
            s += ', name="%s"' % self.name
        return s + ")"

    # -------------------------------------------------------------------------------------------------------------------
    def _world_frame(self):
        """The frame the reference attaches to the twist (PoELink._ets_world, PoERobot.py:42-100): z along the screw axis, origin at
        the point of the axis nearest the base origin, x towards that point (the base x axis when the axis passes through the
        origin) -- built as SE3.OA(o, a) does: n = o x a, o = a x n, unit columns.  One deviation: where the reference's recipe
        degenerates (axis parallel to the chosen x: o = a x n = 0, NaN in the reference) the base y axis is taken instead."""
        ex, ey, ez = np.eye(3)
        if self.isprismatic:
            a, n, t = self.v, ex, np.zeros(3)
        elif self.isrevolute:
            pp = np.cross(self.w, self.v)
            n = ex if np.isclose(np.linalg.norm(pp), 0.0) else pp / np.linalg.norm(pp)
            a, t = self.w, pp
        else:
            n, a, t = ex, ez, self.v
        o = np.cross(a, n)
        if np.linalg.norm(o) < 1e-12 * max(1.0, float(np.linalg.norm(a))):
            o = np.cross(a, ey)
        nn = np.cross(o, a)
        oo = np.cross(a, nn)
        T = np.eye(4)
        T[:3, 0], T[:3, 1], T[:3, 2] = (x / np.linalg.norm(x) for x in (nn, oo, a))
        T[:3, 3] = t
        return T


class PoERevolute(PoELink):
    """PoERevolute(axis, point): rotation about `axis` through `point` (PoERobot.py:124-139; Twist3.UnitRevolute: w = unit axis,
    v = -w x point)."""
    isrevolute = True

    def __init__(self, axis, point, **kwargs):
        w = _unit(axis, "axis")
        p = np.asarray(point, dtype=np.float64).reshape(-1)
        if p.shape != (3,):
            raise ValueError("point must be a 3-vector")
        super().__init__(np.r_[-np.cross(w, p), w], **kwargs)


class PoEPrismatic(PoELink):
    """PoEPrismatic(axis): translation along `axis` (PoERobot.py:142-154; Twist3.UnitPrismatic: w = 0, v = unit axis)."""
    isprismatic = True

    def __init__(self, axis, **kwargs):
        super().__init__(np.r_[_unit(axis, "axis"), np.zeros(3)], **kwargs)


def _rpy_zyx(R):
    """(roll, pitch, yaw) with R = Rz(yaw) Ry(pitch) Rx(roll): spatialmath's tr2rpy(order="zyx") behind SE3.rpy()
    (PoERobot.py:100, :302), including its choice roll = 0 at pitch = +-90 degrees."""
    eps = np.finfo(np.float64).eps
    if abs(abs(R[2, 0]) - 1.0) < 10 * eps:
        yaw = -math.atan2(R[0, 1], R[0, 2]) if R[2, 0] < 0 else math.atan2(-R[0, 1], -R[0, 2])
        return 0.0, -math.asin(min(1.0, max(-1.0, R[2, 0]))), yaw
    roll, yaw = math.atan2(R[2, 1], R[2, 2]), math.atan2(R[1, 0], R[0, 0])
    den = [(R[0, 0], math.cos(yaw)), (R[1, 0], math.sin(yaw)), (R[2, 1], math.sin(roll)), (R[2, 2], math.cos(roll))]
    d, f = max(den, key=lambda x: abs(x[0]))
    return roll, -math.atan(R[2, 0] * f / d), yaw


def _elementary(T):
    """tx ty tz Rz Ry Rx of a transform, the near-zero ones dropped (PoERobot.py:102-112, :304-314)."""
    roll, pitch, yaw = _rpy_zyx(T[:3, :3])
    seq = [ET.tx(float(T[0, 3])), ET.ty(float(T[1, 3])), ET.tz(float(T[2, 3])), ET.Rz(yaw), ET.Ry(pitch), ET.Rx(roll)]
    return [e for e in seq if not np.isclose(e.eta, 0.0)]


def _const_product(ets):
    T = np.eye(4)
    for e in ets:
        T = T @ e.T
    return T


class _PoEChain(ETS):
    """The robot's chain as the kernels run it: the ET list is the reference's (so n, jindices, qlim, repr read as `ets()`),
    the device program is compiled from the twists themselves (rtbhip_chain_create_poe) -- exact to rounding where the
    roll-pitch-yaw re-expression drops elements below 1e-8."""

    def __init__(self, ets, twists, T0):
        super().__init__(ets)
        self._twists = np.ascontiguousarray(twists, dtype=np.float64).reshape(-1, 6)
        self._T0 = np.ascontiguousarray(T0, dtype=np.float64)

    def _handle(self):
        if self._handle_ is None:
            ql = np.ascontiguousarray(self._limits(False).reshape(-1)) if self.n else None
            h = C.c_uint64(0)
            check(lib().rtbhip_chain_create_poe(host_ptr(self._twists), int(self._twists.shape[0]), host_ptr(self._T0), host_ptr(ql), C.byref(h)))
            self._handle_ = h.value
            if self._q_width is not None:
                check(lib().rtbhip_chain_set_q_width(self._handle_, int(self._q_width)))
        return self._handle_


class PoERobot(RobotKinematics):
    """PoERobot(links, T0) (reference PoERobot.py:157-324): `links` PoERevolute / PoEPrismatic in joint order, T0 the
    end-effector pose at q = 0 (4x4 ndarray or an object with `.A`).  fkine / jacob0 / jacobe (and every other method of
    RobotKinematics: hessian0, ik_LM, ikine_LM, manipulability, ...) take one configuration or an (N, n) array / CUDA tensor."""

    def __init__(self, links, T0, name="", base=None, tool=None, **kwargs):
        if kwargs:
            raise TypeError("unexpected keyword argument(s): %s" % ", ".join(sorted(kwargs)))
        joints = list(links)
        for l in joints:
            if not isinstance(l, PoELink) or not l.isjoint:
                raise TypeError("links must be PoERevolute / PoEPrismatic")
        self.T0 = as_se3(T0, "T0").copy()
        if np.any(self.T0[3] != [0, 0, 0, 1]):
            raise ValueError("T0 must be an SE(3) matrix")
        # the reference adds a base link and an end-effector link around the joints (PoERobot.py:174-175)
        self.links = [PoELink(np.zeros(6))] + joints + [PoELink(_se3_log(self.T0))]
        if isinstance(links, list):
            links[:] = self.links                                  # ... in the caller's list, as the reference does (insert / append)
        self.n = len(joints)
        self.name = name
        self.base = as_se3(base, "base")
        self.tool = as_se3(tool, "tool")
        self._link_ets = self._update_ets()
        self._ets_full = None
        self._chain = None

    def __len__(self): return len(self.links)
    def __getitem__(self, i): return self.links[i]
    def __iter__(self): return iter(self.links)
    def nbranches(self): return 0

    def __str__(self):
        s = "PoERobot:\n"
        for j, link in enumerate(self.links):
            s += "  %d: %s\n" % (j, np.array2string(link.S, precision=4, suppress_small=True))
        return s + "  T0: %s" % np.array2string(self.T0, precision=4, suppress_small=True).replace("\n", " ")

    def __repr__(self):
        return "PoERobot([\n" + "\n".join("    %r," % l for l in self.links) + "\n    ],\n    T0=%s,\n    name=\"%s\",\n)" % (
            np.array_repr(self.T0), self.name)

    @property
    def twists(self):
        """(n,6) joint twists (v, w), rows in joint order."""
        return np.array([l.S for l in self.links[1:-1]]).reshape(-1, 6)

    # ------------------------------------------------------------------------------------------------------------------
    def _update_ets(self):
        """PoERobot._update_ets (PoERobot.py:272-324): the world frame of every link AFTER its round trip through the elementary
        list (Link.Ts of the world ETS, :284), the end-effector's replaced by T0 (:286); partial transforms between consecutive
        frames (:292-296), each as tx ty tz Rz Ry Rx without the near-zero elements, the joint appended with jindex i - 1."""
        world = [_const_product(_elementary(l._world_frame())) for l in self.links]
        world[-1] = self.T0
        out = [[]]
        for i in range(1, self.n + 2):
            Wp = world[i - 1]
            inv = np.eye(4)
            inv[:3, :3] = Wp[:3, :3].T
            inv[:3, 3] = -Wp[:3, :3].T @ Wp[:3, 3]
            ets = _elementary(inv @ world[i])
            l = self.links[i]
            if l.isrevolute:
                ets.append(ET.Rz(jindex=i - 1))
            elif l.isprismatic:
                ets.append(ET.tz(jindex=i - 1))
            out.append(ets)
        return out

    def ets(self, start=None, end=None):
        """The robot as elementary transforms (reference BaseRobot.ets on the links `_update_ets` rewrote); start / end are link
        indices (0 = base link .. n + 1 = end-effector link) or PoELink objects, the start link's own transform included as in
        the reference (robot/BaseRobot.py:1426-1467)."""
        def index(x, default):
            if x is None:
                return default
            if isinstance(x, PoELink):
                for k, l in enumerate(self.links):
                    if l is x:
                        return k
                raise ValueError("link not in robot links")
            if isinstance(x, str):
                for k, l in enumerate(self.links):
                    if l.name == x:
                        return k
                raise ValueError("no link named %s" % x)
            k = int(x)
            if not 0 <= k < len(self.links):
                raise ValueError("link not in robot links")
            return k
        a, b = index(start, 0), index(end, len(self.links) - 1)
        if a > b:
            raise ValueError("Could not find the requested ETS in this robot")
        if (a, b) == (0, len(self.links) - 1):
            if self._ets_full is None:
                self._ets_full = ETS([e for seg in self._link_ets for e in seg])
            return self._ets_full
        return ETS([e for seg in self._link_ets[a:b + 1] for e in seg])

    def _path(self, start, end):
        """The whole robot runs on the chain compiled from the twists (the closed form of PoERobot.fkine / jacob0 / jacobe);
        a sub-path is served by the elementary-transform form, as in the reference."""
        if start is None and end is None:
            if self._chain is None:
                self._chain = _PoEChain([e for seg in self._link_ets for e in seg], self.twists, self.T0)
            return self._chain
        return super()._path(start, end)

    @property
    def qlim(self):
        return self._path(None, None).qlim

    @qlim.setter
    def qlim(self, v):
        self._path(None, None).qlim = v

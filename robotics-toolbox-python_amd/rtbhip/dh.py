"""DH robots: the drop-in surface for ``DHRobot.fkine / jacob0 / jacobe / rne``.

Mirrors reference robot/DHLink.py (RevoluteDH, PrismaticDH, RevoluteMDH, PrismaticMDH; `_to_ets`
:173-225) and robot/DHRobot.py (`ets` :878-918, `fkine` :920-979, `_init_rne` :1340-1361,
`rne` :1373-1456, `delete_rne` :1363-1371).  Kinematics run through the ETS lowering on the GPU;
inverse dynamics through rtbhip_rne.  No CPU arithmetic path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib, as_numeric, host_ptr, is_torch, MEM_HOST, MEM_DEVICE
from .linkdyn import LinkDynamics
from .et import ET, ETS, _poses
from .kinematics import RobotKinematics


class DHLink(LinkDynamics):
    def __init__(self, d=0.0, alpha=0.0, theta=0.0, a=0.0, sigma=0, mdh=False, offset=0.0, flip=False,
                 qlim=None, m=None, r=None, I=None, Jm=None, G=None, B=None, Tc=None, name=None, **kw):
        self._robot = None                                    # the DHRobot this link belongs to: told when a dynamic parameter changes
        self.d, self.alpha, self.theta, self.a = float(d), float(alpha), float(theta), float(a)
        self.sigma, self.mdh, self.offset, self.flip = int(sigma), bool(mdh), float(offset), bool(flip)
        self.qlim = None if qlim is None else np.asarray(qlim, dtype=np.float64).reshape(2)
        self.name, self.id, self.number = name, None, None
        self._set_dynamics(m=m, r=r, I=I, Jm=Jm, G=G, B=B, Tc=Tc)

    _DYN = ("m", "r", "I", "Jm", "G", "B", "Tc")
    _KIN = ("d", "a", "alpha", "theta", "offset", "flip", "sigma", "mdh", "qlim")

    def __setattr__(self, name, value):
        """A changed dynamic parameter invalidates the owning robot's device link table, as the reference's setters re-arm `frne.init`
        through `_listen_dyn` (robot/Link.py:28-45 -> DHRobot.dynchanged robot/DHRobot.py:1328-1338)."""
        object.__setattr__(self, name, value)
        if name in DHLink._DYN or name in DHLink._KIN:
            robot = self.__dict__.get("_robot")
            if robot is not None:
                robot.dynchanged()                     # the link record holds alpha, a, theta, d, sigma, offset too (robot/DHRobot.py:1342-1358)
                if name in DHLink._KIN:
                    robot._kinchanged()                # and the kept chains (with their device tables) are rebuilt from the new geometry

    @property
    def isrevolute(self): return self.sigma == 0
    @property
    def isprismatic(self): return self.sigma == 1
    @property
    def isflip(self): return self.flip

    def copy(self):
        """A deep copy that still belongs to the same robot (robot/DHLink.py:390-420 keeps `_robot`)."""
        import copy as _copy
        robot = self.__dict__.get("_robot")
        object.__setattr__(self, "_robot", None)
        try:
            l = _copy.deepcopy(self)
        finally:
            object.__setattr__(self, "_robot", robot)
        object.__setattr__(l, "_robot", robot)
        return l

    def A(self, q):
        """The link transform at joint value q: the closed form of reference robot/DHLink.py:633-673, evaluated on the device as
        the one-link chain `ets()` lowers to.  A `flip`ped joint is flipped here for every link, as it is in
        `ets()` and therefore in the Jacobians of both libraries; the reference's closed form asks the LAST element of the link's ETS
        (robot/DHLink.py:636-639), which for a standard-DH link with a trailing tz(d) / tx(a) / Rx(alpha) is a constant, and so leaves
        the flip out of `A` and `DHRobot.fkine` (a product of `A`s) while its `jacob0` keeps it.  That inconsistency is not reproduced."""
        return self.ets().fkine(np.array([float(q)]))

    def __str__(self):
        """reference robot/DHLink.py:355-376"""
        off = "" if self.offset == 0 else " + %s" % self.offset
        qv = "q" if self.id is None else "q%s" % self.id
        cls = type(self).__name__
        if self.isrevolute:
            return "%s:   θ=%s%s,  d=%s,  a=%s,  ⍺=%s" % (cls, qv, off, self.d, self.a, self.alpha)
        return "%s:  θ=%s,  d=%s%s,  a=%s,  ⍺=%s" % (cls, self.theta, qv, off, self.a, self.alpha)

    def __add__(self, other):
        """link + link / link + robot -> DHRobot (reference robot/DHLink.py:227-255)."""
        if isinstance(other, DHLink):
            return DHRobot([self, other])
        if isinstance(other, DHRobot):
            return DHRobot([self] + list(other.links))
        raise TypeError("Cannot add a DHLink with %s" % type(other).__name__)

    def ets(self):
        """DH -> elementary transforms, same sequence as reference robot/DHLink.py:173-225."""
        out = []
        a, al, th, d, off, fl = self.a, self.alpha, self.theta, self.d, self.offset, self.flip
        if self.mdh:
            if a != 0: out.append(ET.tx(a))
            if al != 0: out.append(ET.Rx(al))
            if self.isrevolute:
                if off != 0: out.append(ET.Rz(off))
                if d != 0: out.append(ET.tz(d))
                out.append(ET.Rz(flip=fl, qlim=self.qlim))
            else:
                if th != 0: out.append(ET.Rz(th))
                if off != 0: out.append(ET.tz(off))
                out.append(ET.tz(flip=fl, qlim=self.qlim))
        else:
            if self.isrevolute:
                if off != 0: out.append(ET.Rz(off))
                out.append(ET.Rz(flip=fl, qlim=self.qlim))
                if d != 0: out.append(ET.tz(d))
            else:
                if th != 0: out.append(ET.Rz(th))
                if off != 0: out.append(ET.tz(off))
                out.append(ET.tz(flip=fl, qlim=self.qlim))
            if a != 0: out.append(ET.tx(a))
            if al != 0: out.append(ET.Rx(al))
        return ETS(out)


class RevoluteDH(DHLink):
    def __init__(self, d=0.0, a=0.0, alpha=0.0, offset=0.0, qlim=None, flip=False, **kw):
        super().__init__(d=d, a=a, alpha=alpha, theta=0.0, sigma=0, mdh=False, offset=offset, qlim=qlim, flip=flip, **kw)


class PrismaticDH(DHLink):
    def __init__(self, theta=0.0, a=0.0, alpha=0.0, offset=0.0, qlim=None, flip=False, **kw):
        super().__init__(theta=theta, a=a, alpha=alpha, d=0.0, sigma=1, mdh=False, offset=offset, qlim=qlim, flip=flip, **kw)


class RevoluteMDH(DHLink):
    def __init__(self, d=0.0, a=0.0, alpha=0.0, offset=0.0, qlim=None, flip=False, **kw):
        super().__init__(d=d, a=a, alpha=alpha, theta=0.0, sigma=0, mdh=True, offset=offset, qlim=qlim, flip=flip, **kw)


class PrismaticMDH(DHLink):
    def __init__(self, theta=0.0, a=0.0, alpha=0.0, offset=0.0, qlim=None, flip=False, **kw):
        super().__init__(theta=theta, a=a, alpha=alpha, d=0.0, sigma=1, mdh=True, offset=offset, qlim=qlim, flip=flip, **kw)


def _mat4(T):
    if T is None:
        return None
    if hasattr(T, "A") and not isinstance(T, np.ndarray):
        T = T.A
    T = np.asarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("expected a 4x4 transform")
    return T.copy()


class DHRobot(RobotKinematics):
    def __init__(self, links, name="", manufacturer="", base=None, tool=None, gravity=None, **kw):
        flat = []
        for l in links:                                       # links, or whole robots whose links are spliced in (robot/DHRobot.py:90-112)
            if isinstance(l, DHLink):
                flat.append(l)
            elif isinstance(l, DHRobot):
                flat.extend(l.links)
            else:
                raise TypeError("Input can be only DHLink or DHRobot")
        self.links = flat
        if not self.links:
            raise ValueError("no links")
        for k, l in enumerate(self.links):
            object.__setattr__(l, "_robot", self)                 # robot/DHRobot.py:114-120: a link reports its parameter changes here
            l.number = k + 1
        if len({l.mdh for l in self.links}) != 1:
            raise ValueError("Robot has mixed D&H links conventions")  # reference robot/DHRobot.py:90-112
        self.name, self.manufacturer = name, manufacturer
        self._ets = None
        self.base = base
        self.tool = tool
        self.gravity = np.array([0.0, 0.0, -9.81]) if gravity is None else np.asarray(gravity, dtype=np.float64).reshape(3)
        self._ets = None
        self._dyn = None
        self.q = np.zeros(len(self.links))                    # the stored configuration (BaseRobot.q): what islimit() etc. default to
        self._control_mode = "v"

    # base and tool live inside ets() (robot/DHRobot.py:878-918 rebuilds that chain on every call; here it is built once and kept with its
    # device table): assigning either one afterwards drops the kept chains, so the next call sees the new transform as the reference's does
    def _kinchanged(self):
        self._ets = None
        self.__dict__.pop("_sub_ets", None)
        self._paths_changed()

    @property
    def base(self): return self._base
    @base.setter
    def base(self, T):
        self._base = _poses(_mat4(T))              # answers to .A / .t / .R like the SE3 the reference hands back
        self._kinchanged()

    @property
    def gravity(self): return self._gravity
    @gravity.setter
    def gravity(self, g):
        self._gravity = np.asarray(g, dtype=np.float64).reshape(3).copy()          # robot/BaseRobot.py:903-906

    @property
    def tool(self): return self._tool_T
    @tool.setter
    def tool(self, T):
        self._tool_T = _poses(_mat4(T))
        self._kinchanged()

    @property
    def control_mode(self): return self._control_mode

    @control_mode.setter
    def control_mode(self, cn):
        if cn not in ("p", "v", "a"):
            raise ValueError("Control type must be one of 'p', 'v', or 'a'")          # robot/BaseRobot.py:1330-1338
        self._control_mode = cn

    def __add__(self, other):
        """robot + robot / robot + link -> a longer DHRobot (reference robot/DHRobot.py:299-327)."""
        if isinstance(other, DHRobot):
            return DHRobot(self.links + other.links, name=self.name + other.name, gravity=self.gravity)
        if isinstance(other, DHLink):
            return DHRobot(self.links + [other], name=self.name, manufacturer=self.manufacturer, gravity=self.gravity)
        raise TypeError("can only add DHRobot or DHLink to DHRobot")

    # per-link parameter vectors (reference robot/DHRobot.py:345-470)
    def _vec(self, name): return [getattr(l, name) for l in self.links]              # lists, as the reference returns them
    @property
    def d(self): return self._vec("d")
    @property
    def a(self): return self._vec("a")
    @property
    def alpha(self): return self._vec("alpha")
    @property
    def theta(self): return self._vec("theta")
    @property
    def offset(self): return self._vec("offset")
    @property
    def r(self):
        """centres of mass, one column per link; a vector for a one-link robot (robot/DHRobot.py:438-460)"""
        R = np.array([l.r for l in self.links]).T
        return R.reshape(-1) if R.shape[1] == 1 else R

    def isrevolute(self, j): return bool(self.links[j].isrevolute)
    def isprismatic(self, j): return bool(self.links[j].isprismatic)
    @property
    def revolutejoints(self): return [bool(l.isrevolute) for l in self.links]
    @property
    def prismaticjoints(self): return [bool(l.isprismatic) for l in self.links]

    def todegrees(self, q=None):
        """revolute joint values in degrees, prismatic ones untouched (reference robot/DHRobot.py:540-568)"""
        q = np.array(self.q if q is None else q, dtype=np.float64)
        k = np.array(self.revolutejoints)
        q[..., k] *= 180.0 / np.pi
        return q

    def toradians(self, q):
        q = np.array(q, dtype=np.float64)
        k = np.array(self.revolutejoints)
        q[..., k] *= np.pi / 180.0
        return q

    def islimit(self, q=None):
        """per joint: outside its limits? (reference robot/DHRobot.py:714-743)"""
        q = self.q if q is None else np.asarray(q, dtype=np.float64).reshape(-1)
        return [l.islimit(qk) for l, qk in zip(self.links, q)]

    def isspherical(self):
        """last three joints form a spherical wrist (reference robot/DHRobot.py:745-778)"""
        if self.n < 3:
            return False
        L = self.links[self.n - 3:]
        al = (-np.pi / 2, np.pi / 2)
        return bool(L[0].a == 0 and L[1].a == 0 and L[1].d == 0
                    and ((L[0].alpha == al[0] and L[1].alpha == al[1]) or (L[0].alpha == al[1] and L[1].alpha == al[0]))
                    and L[0].sigma == 0 and L[1].sigma == 0 and L[2].sigma == 0)

    def A(self, j, q=None):
        """Product of the link transforms j0 .. j1 (j an index: 0 .. j), without base and tool (reference robot/DHRobot.py:660-712);
        evaluated on the device as the sub-chain of those links."""
        j0, jn = (0, int(j)) if np.isscalar(j) else (int(j[0]), int(j[1]))
        jn += 1
        if jn > self.n:
            raise ValueError("The joints value out of range")
        q = np.asarray(self.q if q is None else q, dtype=np.float64).reshape(-1)
        cache = self.__dict__.setdefault("_sub_ets", {})
        if (j0, jn) not in cache:
            e = ETS()
            for l in self.links[j0:jn]:
                e = e * l.ets()
            cache[(j0, jn)] = e
        return cache[(j0, jn)].fkine(q[j0:jn])

    def payload(self, m, p=None):
        """A point-mass payload at p in the last link's frame: it REPLACES that link's mass and centre of mass
        (reference robot/Dynamics.py:614-651)."""
        last = self.links[-1]
        last.m = float(m)
        last.r = np.zeros(3) if p is None else np.asarray(p, dtype=np.float64).reshape(3)
        if self._dyn is not None:
            lib().rtbhip_dyn_destroy(self._dyn)
        self._dyn = None                                      # the device table is rebuilt from the links

    def friction(self, qd):
        """joint friction torques at the velocities qd (reference robot/Dynamics.py:511-560)"""
        qd = np.asarray(qd, dtype=np.float64).reshape(-1)
        return np.array([l.friction(v) for l, v in zip(self.links, qd)])

    def nofriction(self, coulomb=True, viscous=False):
        """A copy of the robot without Coulomb (and, if asked, viscous) friction (reference robot/Dynamics.py:562-612)"""
        return DHRobot([l.nofriction(coulomb, viscous) for l in self.links], name=self.name, manufacturer=self.manufacturer, base=self.base,
                       tool=self.tool, gravity=self.gravity)

    def __len__(self): return len(self.links)
    def __iter__(self): return iter(self.links)
    def __getitem__(self, i): return self.links[i]

    @property
    def n(self): return len(self.links)
    @property
    def mdh(self): return int(self.links[0].mdh)

    @property
    def qlim(self):
        lo, hi = [], []
        for l in self.links:
            if l.qlim is not None:
                lo.append(l.qlim[0]); hi.append(l.qlim[1])
            elif l.isrevolute:
                lo.append(-np.pi); hi.append(np.pi)
            else:
                lo.append(0.0); hi.append(1.0)
        return np.array([lo, hi])

    def ets(self, start=None, end=None):
        """reference robot/DHRobot.py:878-918 (base / tool become constant SE3 transforms).  The reference's ets(*args, **kwargs)
        ignores its arguments; here a start / end other than None is refused."""
        self._refuse("ets", start=start, end=end)
        if self._ets is None:
            e = ETS()
            if self.base is not None and not np.array_equal(self.base, np.eye(4)):
                e = e * ET.SE3(self.base)
            for l in self.links:
                e = e * l.ets()
            if self.tool is not None and not np.array_equal(self.tool, np.eye(4)):
                e = e * ET.SE3(self.tool)
            self._ets = e
        return self._ets

    # ------------------------------------------------------------ kinematics
    # DHRobot is a Robot in the reference (robot/DHRobot.py:38): apart from the five methods it defines itself (below) it inherits
    # the RobotKinematics surface, every call of which goes through self.ets(start, end) -- and DHRobot.ets takes and ignores any
    # arguments (:878).  A drop-in must not ignore silently: None is accepted, anything else is refused.
    def _kin_base(self): return None        # base and tool are inside ets()
    def _kin_tool(self): return None

    @staticmethod
    def _refuse(method, **kw):
        bad = sorted(k for k, v in kw.items() if v is not None)
        if bad:
            raise TypeError("DHRobot.%s: %s has no effect on a DH robot in the reference and is refused here" % (method, ", ".join(bad)))

    def fkine(self, q, **kwargs):
        """(4,4) or (N,4,4).  The reference multiplies closed-form DH matrices in Python (robot/DHRobot.py:920-979, DHLink.A
        robot/DHLink.py:633-673), base and tool included; the ETS lowering evaluated on the GPU is the same product (tests pin
        the two against each other).  The reference's signature is fkine(q, **kwargs) with the keywords unused."""
        self._refuse("fkine", **kwargs)
        if isinstance(q, (list, tuple, np.ndarray)):
            a = np.asarray(q)
            if a.ndim == 1 and a.size > self.n and a.size % self.n == 0:
                q = a.reshape(-1, self.n)                     # a flat run of configurations (getmatrix(q, (None, n)), robot/DHRobot.py:960)
        return self.ets().fkine(q)

    def fkine_all(self, q=None, old=True):
        """Poses of frames {0} (the base) to {n}: (n+1,4,4) or (N,n+1,4,4).  reference robot/DHRobot.py:1018-1064:
        Tj = base; Tj *= L.A(q_j) for every link -- the tool is not applied; q defaults to the stored configuration."""
        q = self.q if q is None else q
        e = self.ets()
        k = 1 if (self.base is not None and not np.array_equal(self.base, np.eye(4))) else 0
        marks = [k]
        for l in self.links:
            k += len(l.ets())
            marks.append(k)
        return _poses(e.link_frames(q, marks))

    def ikine_LM(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, joint_limits=False, mask=None, seed=None, **batch):
        """The reference's DHRobot overrides RobotKinematics.ikine_LM with this narrower signature -- no start / end, and joint limits
        NOT enforced unless asked (robot/DHRobot.py:2454-2474; `ETS.ikine_LM` and `Robot.ikine_LM` default to True): same positional order,
        same defaults.  Further keywords go to ETS.ikine_LM, which checks them."""
        return self.ets().ikine_LM(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, joint_limits=joint_limits, mask=mask, seed=seed, **batch)

    @staticmethod
    def _half(J, half):
        """robot/DHRobot.py:1131-1139, :1188-1196: the translational or the rotational rows."""
        if half is None:
            return J
        if half == "trans":
            return J[..., :3, :]
        if half == "rot":
            return J[..., 3:, :]
        raise ValueError("bad half specified")

    def jacobe(self, q, half=None, **kwargs):
        """Jacobian in the end-effector frame (robot/DHRobot.py:1066-1140); half = "trans" / "rot" returns those three rows."""
        self._refuse("jacobe", **kwargs)
        return self._half(self.ets().jacobe(q), half)

    def _rotate_rows(self, Je, T):
        """tr2jac(T) @ Je (robot/DHRobot.py:1182): both halves of every column rotated by T's rotation -- the one place a supplied
        pose enters; a pre-multiplication of what the kernel returned."""
        T = T.A if hasattr(T, "A") and not isinstance(T, np.ndarray) and not is_torch(T) else T
        if is_torch(Je):
            import torch
            R = torch.as_tensor(T, dtype=Je.dtype, device=Je.device)[..., :3, :3]
            return torch.cat([R @ Je[..., :3, :], R @ Je[..., 3:, :]], dim=-2)
        R = np.asarray(T, dtype=np.float64)[..., :3, :3]
        return np.concatenate([R @ Je[..., :3, :], R @ Je[..., 3:, :]], axis=-2)

    def jacob0(self, q=None, T=None, half=None, start=None, end=None):
        """World-frame Jacobian (robot/DHRobot.py:1142-1198): tr2jac(T) @ jacobe(q) with T = fkine(q) unless the caller supplies
        it.  Without T this is the kernel's jacob0 directly (the same matrix); with T the end-effector-frame Jacobian is rotated by
        the SUPPLIED pose, as the reference does."""
        self._refuse("jacob0", start=start, end=end)
        if T is None:
            return self._half(self.ets().jacob0(q), half)
        return self._half(self._rotate_rows(self.ets().jacobe(q), T), half)

    def jacob0_analytical(self, q, representation=None, T=None):
        """robot/DHRobot.py:1200-1275: representation None returns jacob0."""
        if representation is None:
            return self.jacob0(q, T=T)
        if T is not None:
            raise TypeError("DHRobot.jacob0_analytical: a supplied pose T is not offered with a representation (the kernel forms the "
                            "rate transform from the pose it computes)")
        return self.ets().jacob0_analytical(q, representation=representation)

    def hessian0(self, q=None, J0=None, start=None, end=None):
        """robot/DHRobot.py:1277-1331: self.ets().hessian0(q, J0)."""
        self._refuse("hessian0", start=start, end=end)
        return self.ets().hessian0(q, J0=J0)

    # hessiane, partial_fkine0, manipulability, jacobm, jacob0_dot, ik_LM / ik_GN / ik_NR, ikine_LM / NR / GN / QP: RobotKinematics

    # ------------------------------------------------------------ dynamics
    def L24(self):
        """The 24-double/link block of reference robot/DHRobot.py:1342-1358."""
        L = np.zeros((self.n, 24))
        for i, l in enumerate(self.links):
            L[i, 0:6] = [l.alpha, l.a, l.theta, l.d, l.sigma, l.offset]
            L[i, 6] = l.m
            L[i, 7:10] = l.r
            L[i, 10:19] = l.I.flatten()
            L[i, 19:24] = [l.Jm, l.G, l.B, l.Tc[0], l.Tc[1]]
        return np.ascontiguousarray(L)

    def dynchanged(self, what=None):
        self.delete_rne()

    def delete_rne(self):
        if self._dyn is not None and _lib._lib is not None:
            _lib._lib.rtbhip_dyn_destroy(self._dyn)
        self._dyn = None

    def __getstate__(self):
        """copy.copy / copy.deepcopy / pickle: the device link table and the kept chains belong to THIS robot; a copy builds its own."""
        state = dict(self.__dict__)
        state["_dyn"], state["_ets"] = None, None
        state.pop("_sub_ets", None)
        state.pop("_path_cache", None)
        return state

    def __del__(self):
        try:
            self.delete_rne()
        except Exception:
            pass

    def _dyn_handle(self):
        if self._dyn is None:
            L = self.L24()
            h = C.c_uint64(0)
            check(lib().rtbhip_dyn_create(host_ptr(L), self.n, self.mdh, C.byref(h)))
            self._dyn = h.value
        return self._dyn

    def upload(self, device=None):
        """Chain and link tables resident on `device` now (rtbhip_chain_upload / rtbhip_dyn_upload): hipGraph capture needs no warm-up."""
        dev = -1 if device is None else int(getattr(device, "index", device) or 0)
        self.ets().upload(device)
        check(lib().rtbhip_dyn_upload(self._dyn_handle(), dev))
        return self

    def _dyn_args(self, arrays):
        """-> (list of (N,n) contiguous arrays or None, N, single, torch_mode, ptr(), stream, mem, device)"""
        n = self.n
        first = next(x for x in arrays if x is not None)
        tm = is_torch(first) and first.is_cuda
        out = []
        if tm:
            single = first.dim() == 1
            for x in arrays:
                out.append(None if x is None else x.reshape(-1, n).contiguous())
        else:
            single = as_numeric(first).ndim == 1
            for x in arrays:
                out.append(None if x is None else np.ascontiguousarray(as_numeric(x).reshape(-1, n)))
        N = next(x for x in out if x is not None).shape[0]
        single = single or N == 1                # reference returns row 0 whenever trajn == 1
        if any(x is not None and tuple(x.shape) != (N, n) for x in out):
            raise ValueError("all inputs must be (%d,) or (N,%d)" % (n, n))
        if tm:
            _lib.note_device(first)
            return out, N, single, True, (lambda x: None if x is None else C.c_void_p(x.data_ptr())), \
                _lib.current_stream_ptr(), MEM_DEVICE, first.device
        return out, N, single, False, host_ptr, None, MEM_HOST, None

    @staticmethod
    def _empty(shape, tm, device):
        if tm:
            import torch
            return torch.empty(shape, dtype=torch.float64, device=device)
        return _lib.host_empty(shape)

    def _gravity_c(self, gravity):
        g = self.gravity if gravity is None else np.asarray(gravity, dtype=np.float64).reshape(3)
        if self.base is not None:
            g = self.base[:3, :3].T @ g          # reference robot/DHRobot.py:1431-1433
        return np.ascontiguousarray(-g)          # "we negate gravity here" robot/DHRobot.py:1449

    def rne(self, q, qd=None, qdd=None, gravity=None, fext=None, base_wrench=False):
        """Inverse dynamics tau(q, qd, qdd): (n,) or (N,n)
        (reference robot/DHRobot.py:1373-1456 -> frne.frne core/frne.c:106-230).
        qd / qdd = None means zeros (no zero arrays are read by the kernel)."""
        if base_wrench:
            # robot/DHRobot.py:1409-1412 sends base_wrench=True to rne_python (:1458-1796), which returns (tau, wbase) with
            # wbase = [R_1 f_1, R_1 n_1] (:1765-1770): the wrench the base exerts on link 1 as the backward recursion holds it when it ends,
            # turned into the robot's frame 0.  That second, pure-Python formulation agrees with the compiled frne for standard DH chains
            # without a base transform, and pins this path there (tests/test_06_reference_dh_classes.py).  For a modified-DH chain it does
            # not (:1640 rotates only the first term of the linear acceleration, :1711 takes F_j's moment about p*; the torques it returns
            # differ from DHRobot.rne's) and with a base it enters gravity with the opposite sign (:1597 against :1591).  Those two cases are
            # served by the DEFINITION -- the same wrench from frne's own recursion, with frne's torques and rne's gravity (turned by the
            # base, :1431-1433) -- and checked against momentum balance (tests/test_base_wrench_balance.py).  The moment refers to the origin
            # of frame 0 (standard DH) / of frame 1 (modified DH), as n_1 does.  The reference allocates wbase as (N, n) (:1557), so its call
            # only succeeds for six-joint robots; here wbase is (N, 6) for any n.  A prismatic joint's extension is q + offset as in frne
            # (rne_python drops the offset, :1612).
            arrs, N, single, tm, ptr, stream, mem, dev = self._dyn_args([q, qd, qdd])
            gc = self._gravity_c(gravity)
            f = None if fext is None else np.ascontiguousarray(np.asarray(fext, dtype=np.float64).reshape(6))
            tau = self._empty((N, self.n), tm, dev)
            wb = self._empty((N, 6), tm, dev)
            check(lib().rtbhip_rne_base_wrench(self._dyn_handle(), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), N, host_ptr(gc),
                                               host_ptr(f), ptr(tau), ptr(wb), mem, stream))
            return (tau[0], wb[0]) if single else (tau, wb)
        arrs, N, single, tm, ptr, stream, mem, dev = self._dyn_args([q, qd, qdd])
        gc = self._gravity_c(gravity)
        f = None if fext is None else np.ascontiguousarray(np.asarray(fext, dtype=np.float64).reshape(6))
        tau = self._empty((N, self.n), tm, dev)
        check(lib().rtbhip_rne(self._dyn_handle(), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), N, host_ptr(gc),
                               host_ptr(f), ptr(tau), mem, stream))
        return tau[0] if single else tau

    # ---- the Dynamics-mixin terms (reference robot/Dynamics.py), one fused kernel each
    def gravload(self, q=None, gravity=None):
        """tau_g(q) (reference Dynamics.gravload robot/Dynamics.py:863-922): rne(q, 0, 0)."""
        return self.rne(q, None, None, gravity=gravity)

    def itorque(self, q, qdd):
        """M(q) qdd (reference Dynamics.itorque robot/Dynamics.py:1407-1465): rne(q, 0, qdd) without gravity."""
        return self.rne(q, None, qdd, gravity=[0, 0, 0])

    def inertia(self, q):
        """Joint-space inertia matrix: (n,n) or (N,n,n) (reference Dynamics.inertia robot/Dynamics.py:704-763:
        n rne calls per configuration; here one kernel pass per unit acceleration inside one lane)."""
        arrs, N, single, tm, ptr, stream, mem, dev = self._dyn_args([q])
        M = self._empty((N, self.n, self.n), tm, dev)
        check(lib().rtbhip_inertia(self._dyn_handle(), ptr(arrs[0]), N, ptr(M), mem, stream))
        return M[0] if single else M

    def coriolis(self, q, qd):
        """Coriolis/centripetal matrix C(q, qd): (n,n) or (N,n,n)
        (reference Dynamics.coriolis robot/Dynamics.py:765-861: n + n(n-1)/2 frictionless rne calls; here the same matrix from
        n two-field passes -- column k is the bilinear form B(qd, e_k) of the velocity torque, csrc/dyn_device.h)."""
        arrs, N, single, tm, ptr, stream, mem, dev = self._dyn_args([q, qd])
        Cm = self._empty((N, self.n, self.n), tm, dev)
        check(lib().rtbhip_coriolis(self._dyn_handle(), ptr(arrs[0]), ptr(arrs[1]), N, ptr(Cm), mem, stream))
        return Cm[0] if single else Cm

    def accel(self, q, qd, torque, gravity=None):
        """Forward dynamics qdd = M^-1 (torque - rne(q, qd, 0)): (n,) or (N,n)
        (reference Dynamics.accel robot/Dynamics.py:424-509).  The solve is an LDL^T of the lower triangle of M as the recursion produces
        it.  One case where that matters: for a modified-DH chain whose FIRST joint is prismatic the reference's recursion (core/ne.c) yields a
        non-symmetric M (the first joint's force misses the link masses; `inertia` returns it as the reference does), and the reference's
        accel solves with that full matrix -- the two answers differ there and neither is physical (tests/test_dropin_differential_cpu.py)."""
        arrs, N, single, tm, ptr, stream, mem, dev = self._dyn_args([q, qd, torque])
        gc = self._gravity_c(gravity)
        qdd = self._empty((N, self.n), tm, dev)
        check(lib().rtbhip_accel(self._dyn_handle(), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), N, host_ptr(gc),
                                 ptr(qdd), mem, stream))
        return qdd[0] if single else qdd

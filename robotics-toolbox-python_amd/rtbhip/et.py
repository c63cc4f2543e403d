"""ET / ETS -- the drop-in surface of the batched kinematics path.

Mirrors the call signatures and result shapes of the reference's ``roboticstoolbox.robot.ET.ET``
and ``roboticstoolbox.robot.ETS.ETS`` for the hot path only (constructors, ``eval``, ``fkine``,
``jacob0``, ``jacobe``, ``hessian0``, ``hessiane``, ``ik_LM``, ``ikine_LM``): reference
robot/ET.py:595-935 (constructors), robot/ETS.py:1006-1199 (eval/fkine/jacob0),
:1327-1420 (hessians), :2014-2170 (ik_LM), :2618-2637 (ikine_LM).  Everything is executed by
librtbhip.so on the GPU; there is no Python/NumPy arithmetic path in this module (elementary
constant matrices are the only thing evaluated on the host, once, at construction).

Batch extension: where the reference takes one configuration (jacob0/jacobe/hessian0/ik_LM), a
2-D ``q`` of shape (N, n) [or ``Tep`` of shape (N,4,4)] returns a leading batch axis.
Inputs may be NumPy arrays (host path: staged through the device) or float64 CUDA torch tensors
(zero-copy device path on the current stream; outputs are torch tensors on the same device).
"""
import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import check, lib, as_numeric, small, host_ptr, is_torch, MEM_HOST, MEM_DEVICE

_AXES = {"Rx": 0, "Ry": 1, "Rz": 2, "tx": 3, "ty": 4, "tz": 5}
_METHODS = {"chan": 0, "wampler": 1, "sugihara": 2, "gn": 3, "nr": 4, "qp": 5}


def _elementary(axis, eta):
    """Constant 4x4 of a static elementary transform (what trotx/troty/trotz/transl give a float)."""
    T = np.eye(4)
    k = _AXES[axis]
    if k <= 2:
        c, s = math.cos(eta), math.sin(eta)
        b, d = (k + 1) % 3, (k + 2) % 3
        T[b, b], T[b, d], T[d, b], T[d, d] = c, -s, s, c
    else:
        T[k - 3, 3] = eta
    return T


def _jacobian_batch(J, n, what="J"):
    """(J3, single, torch_mode) of a finished Jacobian (6,n) or a batch (N,6,n): contiguous float64, host array or CUDA tensor."""
    tm = is_torch(J) and J.is_cuda
    if tm:
        single = J.dim() == 2
        J3 = (J.reshape((1,) + tuple(J.shape)) if single else J).contiguous()
        _lib.note_device(J3)
    else:
        a = as_numeric(J.detach().numpy() if is_torch(J) else J, what)
        single = a.ndim == 2
        J3 = np.ascontiguousarray(a.reshape((1,) + a.shape) if single else a)
    if len(J3.shape) != 3 or J3.shape[1] != 6 or (n is not None and J3.shape[2] != n):
        raise ValueError("%s must be (6,n) or (N,6,n)" % what)
    return J3, single, tm


_MANIP_METHODS = {"yoshikawa": 0, "minsingular": 1, "invcondition": 2}


def manipulability_from_jacobian(J, method="yoshikawa", axes="all", n=None):
    """Robot.manipulability(J=...) (robot/Robot.py:701-905: `if J is not None: w = [mfunc(self, J, q, axes_list)]` :896) of a finished
    Jacobian J (6,n) -> scalar, or of a batch (N,6,n) -> (N,): "yoshikawa" sqrt|det(J_a J_a^T)| (|det J_a| when square), "minsingular",
    "invcondition"; `axes` "all" / "trans" / "rot" or six booleans.  A pure function of J: no chain, no q (rtbhip_manipulability_from_jacobian)."""
    if method not in _MANIP_METHODS:
        raise ValueError("Invalid method chosen")
    mask = ETS._axes_mask(axes)
    J3, single, tm = _jacobian_batch(J, n)
    N, _, nj = J3.shape
    m = ETS._out((N,), J3, tm)
    check(lib().rtbhip_manipulability_from_jacobian(ETS._ptr(J3, tm), N, nj, mask, _MANIP_METHODS[method], ETS._ptr(m, tm),
                                                    MEM_DEVICE if tm else MEM_HOST, ETS._stream(tm)))
    return float(m[0]) if single else m


def jacobm_from_jacobian(J, H=None, axes="all", n=None):
    """Robot.jacobm(J=..., H=...) (robot/Robot.py:1101-1235): the manipulability Jacobian of a finished Jacobian J (6,n) -> (n,1), or of a
    batch (N,6,n) -> (N,n).  H (n,6,n) / (N,n,6,n) is the caller's Hessian; None forms hessian0(J0=J) on the fly (:1206).  The measure is
    Yoshikawa's on the rows `axes` selects (rtbhip_jacobm_from_jacobian)."""
    mask = ETS._axes_mask(axes)
    J3, single, tm = _jacobian_batch(J, n)
    N, _, nj = J3.shape
    H4 = None
    if H is not None:
        if (is_torch(H) and H.is_cuda) != tm:
            raise ValueError("J and H must both be host arrays or both be CUDA tensors")
        if tm:
            H4 = (H.reshape((1,) + tuple(H.shape)) if H.dim() == 3 else H).contiguous()
        else:
            h = as_numeric(H.detach().numpy() if is_torch(H) else H, "H")
            H4 = np.ascontiguousarray(h.reshape((1,) + h.shape) if h.ndim == 3 else h)
        if tuple(H4.shape) != (N, nj, 6, nj):
            raise ValueError("Hessian must be of shape (n,6,n), one per Jacobian")          # (the reference's message names 6xnxn, Robot.py:1211)
    Jm = ETS._out((N, nj), J3, tm)
    check(lib().rtbhip_jacobm_from_jacobian(ETS._ptr(J3, tm), None if H4 is None else ETS._ptr(H4, tm), N, nj, mask, ETS._ptr(Jm, tm),
                                            MEM_DEVICE if tm else MEM_HOST, ETS._stream(tm)))
    return Jm[0].reshape(nj, 1) if single else Jm


def hessian_from_jacobian(J, n=None):
    """H (n,6,n) [or (N,n,6,n)] from a finished Jacobian J (6,n) [or (N,6,n)]: _ETS_hessian (core/methods.cpp:16-32), the
    part of ETS_hessian0 / ETS_hessiane (core/fknm.cpp:583-783) that runs when the caller supplies J.  NumPy in -> NumPy
    out (staged through the device); float64 CUDA tensor in -> tensor out on the current stream."""
    tm = is_torch(J) and J.is_cuda
    if tm:
        single = J.dim() == 2
        J3 = (J.reshape((1,) + tuple(J.shape)) if single else J).contiguous()
        _lib.note_device(J3)
    else:
        a = as_numeric(J.detach().numpy() if is_torch(J) else J, "J")
        single = a.ndim == 2
        J3 = np.ascontiguousarray(a.reshape((1,) + a.shape) if single else a)
    if len(J3.shape) != 3 or J3.shape[1] != 6 or (n is not None and J3.shape[2] != n):
        raise ValueError("J must be (6,n) or (N,6,n)")
    N, _, nj = J3.shape
    H = ETS._out((N, nj, 6, nj), J3, tm)
    check(lib().rtbhip_hessian_from_jacobian(ETS._ptr(J3, tm), N, nj, ETS._ptr(H, tm), MEM_DEVICE if tm else MEM_HOST, ETS._stream(tm)))
    return H[0] if single else H


def _pose_error(Te, Tep, method):
    """(N,6) / (6,) error vectors of rtbhip_p_servo_error: method 0 angle-axis, 1 rpy."""
    tm = is_torch(Te) and Te.is_cuda and is_torch(Tep) and Tep.is_cuda
    def shape(T):
        if tm:
            _lib.note_device(T)
            return T.reshape(-1, 4, 4).contiguous(), T.dim() == 2
        if hasattr(T, "A") and not isinstance(T, np.ndarray) and not is_torch(T):
            T = T.A
        a = as_numeric(T.detach().cpu().numpy() if is_torch(T) else T, "T")
        if a.shape[-2:] != (4, 4):
            raise ValueError("poses must be 4x4")
        return np.ascontiguousarray(a.reshape(-1, 4, 4)), a.ndim == 2
    A, sa = shape(Te)
    B, sb = shape(Tep)
    N = max(A.shape[0], B.shape[0])
    if (A.shape[0] not in (1, N)) or (B.shape[0] not in (1, N)):
        raise ValueError("Te and Tep must hold the same number of poses, or one of them a single pose")
    e = ETS._out((N, 6), A, tm)
    if method == 0:
        check(lib().rtbhip_angle_axis(ETS._ptr(A, tm), A.shape[0], ETS._ptr(B, tm), B.shape[0], ETS._ptr(e, tm),
                                      MEM_DEVICE if tm else MEM_HOST, ETS._stream(tm)))
    else:
        check(lib().rtbhip_p_servo_error(ETS._ptr(A, tm), A.shape[0], ETS._ptr(B, tm), B.shape[0], int(method), ETS._ptr(e, tm),
                                         MEM_DEVICE if tm else MEM_HOST, ETS._stream(tm)))
    return e[0] if (sa and sb) else e


def angle_axis(T, Td):
    """Pose error e = [Td.t - T.t ; angle-axis vector of Td.R T.R^T] (reference `rtb.angle_axis(T, Td)`, tools/p_servo.py:13-20
    -> fknm.Angle_Axis core/fknm.cpp:112-162 -> _angle_axis core/ik.cpp:241-286): (6,) for one pair, (N,6) when either
    argument is a stack of N poses (a single pose on the other side is used for every pair)."""
    return _pose_error(T, Td, 0)


angle_axis_python = angle_axis      # the reference keeps a pure-Python twin for symbolic input (tools/p_servo.py:23-43); one kernel serves both names here


def p_servo(wTe, wTep, gain=1.0, threshold=0.1, method="rpy"):
    """Position-based servoing, batched (reference tools/p_servo.py:46-117): v = diag(gain) e, arrived = sum|e| < threshold, with
    e the error seen from the end-effector frame for method "rpy" (the reference's default: [t ; tr2rpy(order "zyx")] of
    inv(wTe) wTep) or the base-frame angle-axis error for any other method string (p_servo.py:98-99 takes that branch for
    everything that is not "rpy").  (6,), bool for one pair; (N,6), (N,) bools for stacks of poses."""
    k = np.asarray(gain, dtype=np.float64)
    if k.ndim not in (0, 1) or (k.ndim == 1 and k.shape[0] != 6):
        raise ValueError("gain must be a scalar or a 6-vector")
    k6 = np.ascontiguousarray(np.broadcast_to(k, (6,)), dtype=np.float64)
    # one launch (rtbhip_p_servo): the error vector, the gain and the arrival test -- shapes and broadcasting as _pose_error
    tm = is_torch(wTe) and wTe.is_cuda and is_torch(wTep) and wTep.is_cuda
    def shape(T):
        if tm:
            _lib.note_device(T)
            return T.reshape(-1, 4, 4).contiguous(), T.dim() == 2
        if hasattr(T, "A") and not isinstance(T, np.ndarray) and not is_torch(T):
            T = T.A
        a = as_numeric(T.detach().cpu().numpy() if is_torch(T) else T, "T")
        if a.shape[-2:] != (4, 4):
            raise ValueError("poses must be 4x4")
        return np.ascontiguousarray(a.reshape(-1, 4, 4)), a.ndim == 2
    A, sa = shape(wTe)
    B, sb = shape(wTep)
    N = max(A.shape[0], B.shape[0])
    if (A.shape[0] not in (1, N)) or (B.shape[0] not in (1, N)):
        raise ValueError("Te and Tep must hold the same number of poses, or one of them a single pose")
    v = ETS._out((N, 6), A, tm)
    if tm:
        import torch
        flag = torch.empty((N,), dtype=torch.uint8, device=A.device)
    else:
        flag = np.empty((N,), dtype=np.uint8)
    check(lib().rtbhip_p_servo(ETS._ptr(A, tm), A.shape[0], ETS._ptr(B, tm), B.shape[0], 1 if method == "rpy" else 0, host_ptr(k6), float(threshold),
                               ETS._ptr(v, tm), ETS._ptr(flag, tm), MEM_DEVICE if tm else MEM_HOST, ETS._stream(tm)))
    arrived = flag.view(torch.bool) if tm else flag.astype(bool)
    if sa and sb:
        return v[0], bool(arrived[0])
    return v, arrived


class SE3Array(np.ndarray):
    """What `fkine` returns for host input: the (4,4) / (N,4,4) array itself, answering to the few attributes callers of the reference
    use on the spatialmath.SE3 it returns there -- `.A` (the plain ndarray), `.t`, `.R`, `.inv()`, and a length / iteration over poses
    for a stack.  Arithmetic stays NumPy's (no operator is redefined): compose poses with `@`."""

    @property
    def A(self): return np.asarray(self)
    @property
    def t(self): return np.asarray(self)[..., :3, 3]
    @property
    def R(self): return np.asarray(self)[..., :3, :3]

    def inv(self):
        a = np.asarray(self)
        out = np.zeros_like(a)
        Rt = np.swapaxes(a[..., :3, :3], -1, -2)
        out[..., :3, :3] = Rt
        out[..., :3, 3] = -np.einsum("...ij,...j->...i", Rt, a[..., :3, 3])
        out[..., 3, 3] = 1.0
        return out.view(SE3Array)


def _poses(T):
    return T.view(SE3Array) if isinstance(T, np.ndarray) else T


class ET:
    """One elementary transform (reference robot/ET.py BaseET/ET)."""

    def __init__(self, axis, eta=None, flip=False, jindex=None, qlim=None, T=None, unit="rad"):
        self.axis = axis
        if eta is not None:
            if isinstance(eta, str) or not np.isscalar(eta) or isinstance(eta, complex):
                raise TypeError("Symbolic value")  # symbolic chains stay on the reference's Python path
            eta = float(eta)
            if unit.lower().startswith("deg") and axis[0] == "R":          # robot/ET.py:59 (any spelling that starts with "deg")
                eta = eta * math.pi / 180.0
        self.eta = eta
        self.isflip = bool(flip)
        self.fknm = object()          # identity token: shared by copy(), fresh on deepcopy() -- what the reference's C handle is to its tests
        self._jindex = None
        self.jindex = jindex
        self.qlim = None if qlim is None else np.asarray(qlim, dtype=np.float64).reshape(2)
        if axis == "SE3":
            T = np.asarray(T.A if hasattr(T, "A") and not isinstance(T, np.ndarray) else T, dtype=np.float64)
            if T.shape != (4, 4):
                raise ValueError("ET.SE3 needs a 4x4 matrix")
            self.T = T.copy()
            self.isjoint = False
        elif eta is None:
            self.T = np.eye(4)
            self.isjoint = True
        else:
            self.T = _elementary(axis, eta)
            self.isjoint = False

    # constructors, same names/arguments as reference robot/ET.py:611-935
    @classmethod
    def Rx(cls, eta=None, unit="rad", **kw): return cls("Rx", eta, unit=unit, **kw)
    @classmethod
    def Ry(cls, eta=None, unit="rad", **kw): return cls("Ry", eta, unit=unit, **kw)
    @classmethod
    def Rz(cls, eta=None, unit="rad", **kw): return cls("Rz", eta, unit=unit, **kw)
    @classmethod
    def tx(cls, eta=None, **kw): return cls("tx", eta, **kw)
    @classmethod
    def ty(cls, eta=None, **kw): return cls("ty", eta, **kw)
    @classmethod
    def tz(cls, eta=None, **kw): return cls("tz", eta, **kw)
    @classmethod
    def SE3(cls, T, **kw): return cls("SE3", T=T, **kw)

    @property
    def isrotation(self): return self.axis[0] == "R"
    @property
    def istranslation(self): return self.axis[0] == "t"
    @property
    def iselementary(self):
        """False for a general constant (ET.SE3), True for the six axis transforms (reference robot/ET.py:353-368)."""
        return self.axis[0] != "S"

    @property
    def jindex(self): return self._jindex

    @jindex.setter
    def jindex(self, j):
        if j is not None and (not isinstance(j, (int, np.integer)) or isinstance(j, bool) or j < 0):
            raise ValueError("jindex is %r, must be an int >= 0" % (j,))          # robot/ET.py:496-501
        self._jindex = None if j is None else int(j)

    def A(self, q=0.0):
        """The 4x4 of this transform at joint value q (reference ET.A robot/ET.py:560-583 -> fknm.ET_T core/fknm.cpp:1241-1281):
        a constant returns a copy of its matrix; a joint is evaluated on the device as a one-element chain, `flip` included."""
        if not self.isjoint:
            return self.T.copy()
        if isinstance(q, (list, tuple, np.ndarray)):
            q = np.asarray(q, dtype=np.float64).reshape(-1)[0]
        if not isinstance(q, (int, float, np.integer, np.floating)):
            raise TypeError("Symbolic value")
        one = ETS(ET(self.axis, flip=self.isflip, qlim=self.qlim))
        return one.eval(np.array([float(q)]))

    def __mul__(self, other): return ETS(self) * other
    def __add__(self, other): return ETS(self) * other
    def __eq__(self, other): return isinstance(other, ET) and repr(self) == repr(other)      # robot/ET.py:241-242
    __hash__ = None

    def __str__(self):
        """The reference's short form (robot/ET.py:160-198): `Rx(88.41°)`, `tx(1.543)`, `tz(q3)`, `Rx(q)`; a general constant as its
        translation and roll-pitch-yaw angles."""
        if self.isjoint:
            arg = "q" if self.jindex is None else "q%d" % self.jindex
        elif self.axis == "SE3":
            T = self.T
            # roll-pitch-yaw, zyx order (spatialmath tr2rpy default), in degrees -- the singular case (pitch = +-90 degrees) as there
            from .poe import _rpy_zyx
            r, pch, y = _rpy_zyx(T[:3, :3])
            rpy = np.array([r, pch, y]) * 180.0 / math.pi
            t = T[:3, 3]
            ts = "%.4g, %.4g, %.4g" % tuple(t)
            rs = "%.4g°, %.4g°, %.4g°" % tuple(rpy)
            arg = (ts + "; " + rs) if (t.any() and rpy.any()) else (ts if t.any() else (rs if rpy.any() else ""))
        elif self.isrotation:
            arg = "%.4g°" % (self.eta * 180.0 / math.pi)
        else:
            arg = "%.4g" % self.eta
        return "%s(%s)" % (self.axis, arg)

    def inv(self):
        """Inverse of this ET (reference robot/ET.py:413-445): a joint keeps its axis and toggles `flip`, a constant is
        the inverse matrix."""
        if self.isjoint:
            return ET(self.axis, flip=not self.isflip, jindex=self.jindex, qlim=self.qlim)
        if self.axis == "SE3":
            return ET("SE3", T=np.linalg.inv(self.T))
        return ET(self.axis, -self.eta)

    def __repr__(self):
        """`ET.Rx(eta=1.543, jindex=5, flip=True, qlim=array([-1.,  1.]))` (reference robot/ET.py:200-217)."""
        parts = ["" if self.eta is None else "eta=%s" % self.eta,
                 "T=%r" % self.T if self.axis == "SE3" else "",
                 "" if self.jindex is None else "jindex=%d" % self.jindex,
                 "" if not self.isflip else "flip=True",
                 "" if self.qlim is None else "qlim=%r" % self.qlim]
        return "ET.%s(%s)" % (self.axis, ", ".join(x for x in parts if x))

    def _short(self):
        if self.axis == "SE3":
            return "SE3(...)"
        arg = "q%s" % ("" if self.jindex is None else self.jindex) if self.isjoint else "%.4g" % self.eta
        return "%s(%s%s)" % (self.axis, "-" if self.isflip else "", arg)


@dataclass
class IKSolution:
    """Result of ikine_LM (reference robot/IK.py:27-101)."""
    q: np.ndarray
    success: bool = False
    iterations: int = 0
    searches: int = 0
    residual: float = 0.0
    reason: str = ""
    each: dict = field(default_factory=dict, repr=False)  # per-target arrays for a batch of Tep

    def __iter__(self):
        return iter((self.q, self.success, self.iterations, self.searches, self.residual, self.reason))

    def __str__(self):
        """The reference's one-line form (robot/IK.py:60-100): analytic solutions (no iterations, no searches) print without the counters."""
        if self.q is not None:
            q_str = np.array2string(np.asarray(self.q), separator=", ", formatter={"float": lambda x: "{:.4g}".format(0 if abs(x) < 1e-6 else x)})
        else:
            q_str = None
        if self.iterations == 0 and self.searches == 0:
            return "IKSolution: q=%s, success=True" % q_str if self.success else "IKSolution: q=%s, success=False, reason=%s" % (q_str, self.reason)
        if self.success:
            return "IKSolution: q=%s, success=True, iterations=%d, searches=%d, residual=%.3g" % (q_str, self.iterations, self.searches, self.residual)
        return "IKSolution: q=%s, success=False, reason=%s, iterations=%d, searches=%d, residual=%.3g" % (
            q_str, self.reason, self.iterations, self.searches, np.round(self.residual, 4))


class ETS:
    """A sequence of elementary transforms bound to a device chain handle (reference robot/ETS.py)."""

    def __init__(self, arg=None):
        if arg is None:
            ets = []
        elif isinstance(arg, ET):
            ets = [arg]
        elif isinstance(arg, ETS):
            ets = list(arg._ets)
        elif isinstance(arg, (list, tuple)):
            ets = []
            for a in arg:
                if not isinstance(a, (ET, ETS)):
                    raise TypeError("bad arg")                # robot/ETS.py:785-797
                ets.extend(a._ets if isinstance(a, ETS) else [a])
        else:
            raise TypeError("Invalid arg")
        self._ets = ets
        self._handle_ = None
        self._qlim = None
        self._q_width = None

    # ------------------------------------------------------------ structure
    def __mul__(self, other):
        if isinstance(other, ET):
            return ETS(self._ets + [other])
        if isinstance(other, ETS):
            return ETS(self._ets + other._ets)
        return NotImplemented

    __add__ = __mul__

    def inv(self):
        """Inverse ETS: the inverses of the elements in reverse order (reference robot/ETS.py:545-580).  Every joint KEEPS its joint number --
        `ets.inv().eval(q)` is the inverse of `ets.eval(q)` for the same q -- so joints numbered automatically (in order of appearance) carry that
        number explicitly in the inverse, where they appear in the opposite order."""
        idx = iter(self._assigned_jindices())
        numbered = [ET(e.axis, flip=e.isflip, jindex=next(idx), qlim=e.qlim) if e.isjoint else e for e in self._ets]
        return ETS([e.inv() for e in reversed(numbered)])

    def split(self):
        """Link segments: every piece but possibly the last ends with a joint (reference robot/ETS.py:511-543)."""
        out, cur = [], []
        for e in self._ets:
            cur.append(e)
            if e.isjoint:
                out.append(ETS(cur))
                cur = []
        if cur:
            out.append(ETS(cur))
        return out

    def __len__(self): return len(self._ets)
    def __iter__(self): return iter(self._ets)

    def __getitem__(self, i):
        """An ET for an index, an ETS for a slice (reference robot/ETS.py: UserList semantics)."""
        return ETS(self._ets[i]) if isinstance(i, slice) else self._ets[i]

    def __eq__(self, other):
        return isinstance(other, ETS) and len(self) == len(other) and all(a == b for a, b in zip(self._ets, other._ets))
    __hash__ = None

    # ---- list-like editing (the reference's ETS is a UserList, robot/ETS.py:28-60, 430-503); any edit drops the device table
    @property
    def data(self): return self._ets

    def _edited(self):
        self._drop_handle()
        self._qlim = None
        self._q_width = None

    @staticmethod
    def _items(arg):
        if isinstance(arg, ET):
            return [arg]
        if isinstance(arg, ETS):
            return list(arg._ets)
        raise TypeError("can only add / insert an ET or an ETS")

    def copy(self):
        out = ETS(list(self._ets))
        out._qlim = None if self._qlim is None else self._qlim.copy()
        return out

    def append(self, et):
        self._ets.extend(self._items(et)); self._edited()

    def extend(self, ets):
        for e in ets:
            self._ets.extend(self._items(e))
        self._edited()

    def insert(self, arg, i=-1):
        """Insert an ET or an ETS; the inserted value ends up at position i, the default is the end (reference robot/ETS.py:433-470)."""
        items = self._items(arg)
        if i == -1:
            i = len(self._ets)
        for k, e in enumerate(items):
            self._ets.insert(i + k, e)
        self._edited()

    def pop(self, i=-1):
        """Remove and return element i (reference robot/ETS.py:472-503)."""
        item = self._ets.pop(i)
        self._edited()
        return item

    def remove(self, et):
        self._ets.remove(et); self._edited()

    def clear(self):
        self._ets.clear(); self._edited()

    def reverse(self):
        self._ets.reverse(); self._edited()

    def index(self, et, *a): return self._ets.index(et, *a)
    def count(self, et): return self._ets.count(et)

    def joint_idx(self):
        """Positions of the joints within the sequence (reference robot/ETS.py:207-227)."""
        return np.array([i for i, e in enumerate(self._ets) if e.isjoint], dtype=np.int64)

    def jindex_set(self):
        """The set of joint indices (reference robot/ETS.py:249-267)."""
        return set(int(j) for j in self.jindices)

    @property
    def structure(self):
        """'R' / 'P' per joint (reference robot/ETS.py:366-388)."""
        return "".join("R" if self._ets[i].isrotation else "P" for i in self.joint_idx())

    def compile(self):
        """Constants between joints folded into one SE3 each, identities dropped (reference robot/ETS.py:857-906).  (The device chain
        table is compiled this way whether or not this is called: csrc/chain.cpp.)"""
        out, const = [], None
        for e in self._ets:
            if e.isjoint:
                if const is not None and not np.array_equal(const, np.eye(4)):
                    out.append(ET.SE3(const))
                const = None
                out.append(e)
            else:
                const = e.T.copy() if const is None else const @ e.T
        if const is not None and not np.array_equal(const, np.eye(4)):
            out.append(ET.SE3(const))
        return ETS(out)

    def random_q(self, i=1):
        """Uniform samples within the joint limits: (n,) for i = 1, else (i, n) (reference robot/ETS.py:680-724; the reference draws from
        Python's `random.uniform`, here numpy's global generator)."""
        ql = self.qlim
        q = np.random.uniform(ql[0], ql[1], (int(i), self.n))
        return q[0] if i == 1 else q
    def __repr__(self): return " * ".join(e._short() for e in self._ets) or "ETS()"

    def __str__(self, q=None):
        """The reference's one-line form (robot/ETS.py:71-190): `Rx(88.41°) ⊕ Rx(q0) ⊕ tx(1)`; `q` is the format of a joint variable
        ("q{0}" numbered from 0, "θ{1}" numbered from 1; default "q{0}", or plain "q" for a single joint); an empty sequence is `SE3()`."""
        if not self._ets:
            return "SE3()"
        if q is None:
            q = "q{0}" if self.n > 1 else "q"
        jidx = iter(self._assigned_jindices())
        out = []
        for e in self._ets:
            if e.isjoint:
                j = next(jidx)
                out.append("%s(%s%s)" % (e.axis, "-" if e.isflip else "", q.format(j, j + 1)))
            else:
                out.append(str(e))
        return " \u2295 ".join(out)

    def __getstate__(self):
        """copy.copy / copy.deepcopy / pickle: the device chain handle belongs to THIS object (its __del__ destroys it); a copy makes its own."""
        state = dict(self.__dict__)
        state["_handle_"] = None
        return state

    def __del__(self):
        h = getattr(self, "_handle_", None)
        if h is not None and _lib._lib is not None:
            try:
                _lib._lib.rtbhip_chain_destroy(h)
            except Exception:
                pass

    @property
    def m(self): return len(self._ets)

    def joints(self): return [e for e in self._ets if e.isjoint]

    @property
    def n(self): return len(self.joints())

    def _assigned_jindices(self):
        """Joints without an explicit jindex are numbered in order of appearance
        (reference robot/ETS.py:62-100 `_auto_jindex`)."""
        js = self.joints()
        if all(e.jindex is None for e in js):
            return list(range(len(js)))
        if js[-1].jindex is None and all(e.jindex == k for k, e in enumerate(js[:-1])):
            return list(range(len(js)))                       # numbered 0 .. n-2 in order, the last one open: it is n-1 (robot/ETS.py:820-828)
        if any(e.jindex is None for e in js):
            raise ValueError("either all or none of the joints must have a jindex")
        return [int(e.jindex) for e in js]

    @property
    def jindices(self): return np.array(self._assigned_jindices(), dtype=int)

    def _limits(self, strict):
        if self._qlim is not None:
            return self._qlim
        lo, hi = [], []
        for e in self.joints():
            if e.qlim is not None:
                lo.append(e.qlim[0]); hi.append(e.qlim[1])
            elif e.isrotation:
                lo.append(-math.pi); hi.append(math.pi)
            elif strict:
                raise ValueError("undefined prismatic joint limit")       # robot/ETS.py:335-337
            else:
                lo.append(0.0); hi.append(1.0)
        return np.array([lo, hi], dtype=np.float64)

    @property
    def qlim(self):
        """(2, n) joint limits as the reference's ETS.qlim (robot/ETS.py:300-345): an unset revolute limit reads as [-pi, pi], an unset
        prismatic one raises.  (The device table, like the reference's C structs -- robot/ET.py:109-115 -- holds [0, 1] there.)"""
        return self._limits(True)

    @qlim.setter
    def qlim(self, v):
        v = np.asarray(v, dtype=np.float64)
        if v.shape == (2,) and self.n == 1:                 # robot/ETS.py:346-351: (2,) for a single joint, else (2, n) and nothing else
            v = v.reshape(2, 1)
        if v.shape != (2, self.n):
            raise ValueError("new_qlim must be of shape (2, n)")
        self._qlim = np.ascontiguousarray(v)
        self._drop_handle()

    def _drop_handle(self):
        if self._handle_ is not None and _lib._lib is not None:
            _lib._lib.rtbhip_chain_destroy(self._handle_)
        self._handle_ = None

    def optable(self):
        """The flat description handed to rtbhip_chain_create: (kind, flip, jindex, T16) per ET."""
        jidx = iter(self._assigned_jindices())
        rows = []
        for e in self._ets:
            if e.isjoint:
                rows.append((_AXES[e.axis], int(e.isflip), next(jidx), np.eye(4)))
            else:
                rows.append((_lib.ET_CONST, 0, 0, e.T))
        return rows

    def _handle(self):
        if self._handle_ is None:
            rows = self.optable()
            arr = (_lib.rtbhip_et * max(1, len(rows)))()
            for i, (kind, flip, jindex, T) in enumerate(rows):
                arr[i].kind, arr[i].flip, arr[i].jindex, arr[i].reserved = kind, flip, jindex, 0
                flat = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
                for k in range(16):
                    arr[i].T[k] = flat[k]
            ql = np.ascontiguousarray(self._limits(False).reshape(-1)) if self.n else None
            h = C.c_uint64(0)
            check(lib().rtbhip_chain_create(arr, len(rows), host_ptr(ql), C.byref(h)))
            self._handle_ = h.value
            if self._q_width is not None:
                check(lib().rtbhip_chain_set_q_width(self._handle_, int(self._q_width)))
        return self._handle_

    def upload(self, device=None):
        """Make the chain table resident on `device` (None: the current GPU) now (rtbhip_chain_upload): afterwards every call with
        device tensors only enqueues kernels, so a sequence of calls can be captured into a hipGraph without a warm-up call."""
        check(lib().rtbhip_chain_upload(self._handle(), -1 if device is None else int(getattr(device, "index", device) or 0)))
        return self

    @property
    def q_width(self):
        """Columns of a q row: max(jindex)+1, or what was assigned -- a branch of a tree robot reads the robot-wide q
        (reference Robot.jacob0(q, start, end) = self.ets(start, end).jacob0(q) on the ROBOT's q, robot/Robot.py:1974-1981)."""
        j = self._assigned_jindices()
        need = (max(j) + 1) if j else 0
        return need if self._q_width is None else max(need, self._q_width)

    @q_width.setter
    def q_width(self, w):
        j = self._assigned_jindices()
        need = (max(j) + 1) if j else 0
        if w is not None and not need <= int(w) <= 256:
            raise ValueError("q_width must be in [%d, 256] for this chain" % need)
        self._q_width = None if w is None else int(w)
        self._drop_handle()

    # ------------------------------------------------------------ argument shaping
    def _shape_q(self, q):
        """-> (q2d, single, torch_mode).  1-D, (1,n) and (n,1) are ONE configuration
        (reference core/fknm.cpp:964-988); anything else is a trajectory of rows."""
        qw = self.q_width
        if is_torch(q):
            import torch
            if not q.is_cuda:
                return self._shape_q(q.detach().numpy())
            if q.dtype != torch.float64:
                raise TypeError("device q must be float64")
            single = q.dim() == 1 or (q.dim() == 2 and self._one_config(tuple(q.shape), qw))
            q2 = q.reshape(1, -1) if single else q
            if q2.shape[1] > qw:
                q2 = q2[:, :qw]                      # a wider row (the whole robot's q): the joints are read by number, the rest is not looked at
            q2 = q2.contiguous()
            if q2.shape[1] != qw:
                raise ValueError("q has %d columns, chain needs %d" % (q2.shape[1], qw))
            _lib.note_device(q2)
            return q2, single, True
        a = as_numeric(q)
        if a.ndim == 0:
            a = a.reshape(1)
        if a.ndim > 2:
            raise ValueError("q must be 1-D or 2-D")
        single = a.ndim == 1 or self._one_config(a.shape, qw)
        a = a.reshape(1, -1) if single else a
        if a.shape[1] > qw:
            # the reference's C entry points take the row length from the array and read q[jindex] (core/fknm.cpp:964-988, methods.cpp:338): a row
            # wider than the chain needs -- the whole robot's q handed to robot.ets(end=...) -- is read by joint number, the rest is not looked at
            a = a[:, :qw]
        if a.shape[1] != qw:
            raise ValueError("q has %d columns, chain needs %d" % (a.shape[1], qw))
        return np.ascontiguousarray(a), single, False

    @staticmethod
    def _one_config(shape, qw):
        """(1,n) and (n,1) are one configuration (core/fknm.cpp:968-981).  For a one-joint chain the
        reference's rule would swallow every (N,1) trajectory; there (N,1) with N > 1 is a batch."""
        if shape[0] == 1:
            return True
        return shape[1] == 1 and qw != 1

    @staticmethod
    def _out(shape, like, torch_mode, dtype=None):
        if torch_mode:
            import torch
            return torch.empty(shape, dtype=dtype or torch.float64, device=like.device)
        return _lib.host_empty(shape, dtype or np.float64)      # pinned from 1 MB up: the D2H DMA writes into it directly

    @staticmethod
    def _ptr(x, torch_mode):
        if x is None:
            return None
        return C.c_void_p(x.data_ptr()) if torch_mode else host_ptr(x)

    @staticmethod
    def _stream(torch_mode):
        return _lib.current_stream_ptr() if torch_mode else None

    # ------------------------------------------------------------ kinematics
    def eval(self, q, base=None, tool=None, include_base=True):
        """Forward kinematics as ndarray: (4,4) for one q, (N,4,4) for a trajectory
        (reference ETS.eval robot/ETS.py:1021-1141 -> ETS_fkine core/fknm.cpp:923-1064)."""
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        T = self._out((N, 4, 4), q2, tm)
        b = small(base, 16) if (base is not None and include_base) else None
        t = small(tool, 16)
        check(lib().rtbhip_fkine(self._handle(), self._ptr(q2, tm), N, host_ptr(b), host_ptr(t),
                                 self._ptr(T, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return T[0] if single else T

    def link_frames(self, q, marks, base=None):
        """Poses of intermediate frames: frame m = base * (product of the first marks[m] transforms).  (nmarks,4,4) for
        one q, (N,nmarks,4,4) for a trajectory -- what DHRobot.fkine_all / Robot.fkine_all build link by link in Python
        (robot/DHRobot.py:1012-1064, robot/Robot.py:638-698), in one chain walk per configuration."""
        marks = np.ascontiguousarray(marks, dtype=np.int32).reshape(-1)
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        out = self._out((N, len(marks), 4, 4), q2, tm)
        check(lib().rtbhip_link_frames(self._handle(), self._ptr(q2, tm), N, host_ptr(small(base, 16)), host_ptr(marks),
                                       len(marks), self._ptr(out, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return out[0] if single else out

    def fkine(self, q, base=None, tool=None, include_base=True):
        """reference ETS.fkine (robot/ETS.py:1006-1019) wraps eval() in spatialmath.SE3; spatialmath is not a dependency here: host
        input returns the array as an SE3Array (an ndarray that also answers `.A`, `.t`, `.R`, `.inv()`), device input a tensor."""
        return _poses(self.eval(q, base=base, tool=tool, include_base=include_base))

    def _jac(self, q, tool, frame):
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        J = self._out((N, 6, self.n), q2, tm)
        t = small(tool, 16)
        check(lib().rtbhip_jacob(self._handle(), self._ptr(q2, tm), N, host_ptr(t), frame,
                                 self._ptr(J, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return J[0] if single else J

    def jacob0(self, q, tool=None):
        """Geometric Jacobian in the chain's start frame: (6,n), or (N,6,n) for a trajectory
        (reference ETS.jacob0 robot/ETS.py:1143-1199 -> ETS_jacob0 core/fknm.cpp:785-850)."""
        return self._jac(q, tool, 0)

    def jacobe(self, q, tool=None):
        """End-effector-frame Jacobian (reference robot/ETS.py:1274-1330, core/fknm.cpp:852-921)."""
        return self._jac(q, tool, 1)

    def fkine_jacob0(self, q, base=None, tool=None, frame=0, packed=False, out=None):
        """Fused fkine + Jacobian in one chain walk (the headline op; no reference equivalent --
        a reference user calls eval() then jacob0() per row).
        packed=True: ONE (N, 16 + 6n) array whose row i is [T[i] (16, row-major 4x4) | J[i] ((6,n) C-order)] -- the device writes a single
        stream and the row is the T||J message of the multi-GPU gather (rtbhip_fkine_jacob_packed); the returned (T, J) are strided views
        of it and `.packed` / the third item gives the array itself: `T, J, TJ = ets.fkine_jacob0(q, packed=True)`.  `out` = a TJ array of an
        earlier call to write into."""
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        if packed:
            w = 16 + 6 * self.n
            TJ = self._out((N, w), q2, tm) if out is None else out
            if out is not None:
                # the kernel (or the D2H copy) writes N * w * 8 bytes through this pointer: shape, width, contiguity, element type AND where
                # it lives must be what q's are -- a float32 or wrong-device `out` would be overrun / written through a foreign pointer
                import torch as _torch
                good = (is_torch(TJ) and TJ.dtype == _torch.float64 and TJ.device == q2.device and TJ.is_contiguous()) if tm else \
                       (isinstance(TJ, np.ndarray) and TJ.dtype == np.float64 and TJ.flags.c_contiguous and TJ.flags.writeable)
                if not good or tuple(TJ.shape) != (N, w):
                    raise ValueError("out must be a contiguous (N, 16 + 6n) float64 %s" % ("tensor on q's device" if tm else "ndarray"))
            b, t = small(base, 16), small(tool, 16)
            check(lib().rtbhip_fkine_jacob_packed(self._handle(), self._ptr(q2, tm), N, host_ptr(b), host_ptr(t), frame,
                                                  self._ptr(TJ, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
            if tm:
                T, J = TJ[:, :16].unflatten(1, (4, 4)), TJ[:, 16:].unflatten(1, (6, self.n))       # strided views, no copy
            else:
                T = np.lib.stride_tricks.as_strided(TJ, (N, 4, 4), (8 * w, 32, 8))
                J = np.lib.stride_tricks.as_strided(TJ[:, 16:], (N, 6, self.n), (8 * w, 8 * self.n, 8))
            return (T[0], J[0], TJ[0]) if single else (T, J, TJ)
        T = self._out((N, 4, 4), q2, tm)
        J = self._out((N, 6, self.n), q2, tm)
        b, t = small(base, 16), small(tool, 16)
        check(lib().rtbhip_fkine_jacob(self._handle(), self._ptr(q2, tm), N, host_ptr(b), host_ptr(t), frame,
                                       self._ptr(T, tm), self._ptr(J, tm), MEM_DEVICE if tm else MEM_HOST,
                                       self._stream(tm)))
        return (T[0], J[0]) if single else (T, J)

    def _hess(self, q, tool, frame, J=None):
        if J is not None:
            # the reference's own calling form: ETS_hessian0(ets, q, J0, tool) with J0 given only runs _ETS_hessian on it
            # (core/fknm.cpp:583-783 -> methods.cpp:16-32); q and tool are then not looked at
            return hessian_from_jacobian(J, self.n)
        if q is None:
            raise ValueError("one of q or the Jacobian must be supplied")
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        n = self.n
        H = self._out((N, n, 6, n), q2, tm)
        t = small(tool, 16)
        check(lib().rtbhip_hessian(self._handle(), self._ptr(q2, tm), N, host_ptr(t), frame,
                                   self._ptr(H, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return H[0] if single else H

    def hessian0(self, q=None, J0=None, tool=None):
        """(n,6,n) Hessian in the start frame (reference robot/ETS.py:1332-1420, fknm.cpp:583-682); from q, or -- as the
        reference allows -- from an already computed J0 ((6,n), or (N,6,n) for a batch)."""
        return self._hess(q, tool, 0, J0)

    def hessiane(self, q=None, Je=None, tool=None):
        """(n,6,n) Hessian in the end-effector frame (reference robot/ETS.py:1422-1510, fknm.cpp:684-783), from q or Je."""
        return self._hess(q, tool, 1, Je)

    # ------------------------------------------------------------ differential-kinematics consumers
    def jacob0_dot(self, q, qd, J0=None, representation=None, tool=None):
        """d/dt J0 = hessian0(q) . qd: (6,n) or (N,6,n) (reference Robot.jacob0_dot robot/Robot.py:964-1098).  With an
        orientation `representation` ("rpy/xyz", "rpy/zyx", "eul", "exp") the rate of the ANALYTICAL Jacobian, which the
        reference obtains from a forward-difference numerical Hessian of jacob0_analytical (:1090-1093); reproduced as such."""
        return self._jdot(q, qd, representation, tool, 0)

    def jacobe_dot(self, q, qd, tool=None):
        return self._jdot(q, qd, None, tool, 1)

    def _jdot(self, q, qd, representation, tool, frame):
        if representation is not None and representation not in self._REPRESENTATIONS:
            raise ValueError("representation must be one of %s" % ", ".join(self._REPRESENTATIONS))
        q2, single, tm = self._shape_q(q)
        qd2, _, tm2 = self._shape_q(qd)
        if tm != tm2 or tuple(q2.shape) != tuple(qd2.shape):
            raise ValueError("q and qd must have the same shape and live in the same memory")
        N = q2.shape[0]
        Jd = self._out((N, 6, self.n), q2, tm)
        if representation is not None:
            check(lib().rtbhip_jacob0_dot_analytical(self._handle(), self._ptr(q2, tm), self._ptr(qd2, tm), N, host_ptr(small(tool, 16)),
                                                     self._REPRESENTATIONS[representation], self._ptr(Jd, tm),
                                                     MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        else:
            check(lib().rtbhip_jacob_dot(self._handle(), self._ptr(q2, tm), self._ptr(qd2, tm), N, host_ptr(small(tool, 16)), frame,
                                         self._ptr(Jd, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return Jd[0] if single else Jd

    @staticmethod
    def _axes_mask(axes):
        if isinstance(axes, str):
            if axes.startswith("all"):
                return 63
            if axes.startswith("trans"):
                return 7
            if axes.startswith("rot"):
                return 56
            raise ValueError("axes must be all, trans or rot")          # robot/ETS.py:1764
        ax = [bool(a) for a in axes]
        if len(ax) != 6:
            raise ValueError("axes must be all, trans, rot or a list of 6 booleans")
        return sum(1 << i for i, a in enumerate(ax) if a)

    def manipulability(self, q, method="yoshikawa", axes="all", tool=None):
        """Manipulability measure: scalar or (N,) (reference ETS.manipulability robot/ETS.py:1687-1819):
        "yoshikawa" sqrt|det(J J^T)|, "minsingular" the smallest singular value of J, "invcondition"
        s_min / s_max (singular values from an in-register Jacobi eigen-solve of the Gram matrix)."""
        methods = {"yoshikawa": 0, "minsingular": 1, "invcondition": 2}
        if method not in methods:
            raise ValueError("Invalid method chosen")
        mask = self._axes_mask(axes)
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        m = self._out((N,), q2, tm)
        check(lib().rtbhip_manipulability(self._handle(), self._ptr(q2, tm), N, host_ptr(small(tool, 16)), mask, methods[method],
                                          self._ptr(m, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return float(m[0]) if single else m

    def jacobm(self, q, axes="all", tool=None):
        """Manipulability Jacobian: (n,1) for one q as the reference returns it (robot/ETS.py:1628-1685),
        (N,n) for a trajectory."""
        mask = self._axes_mask(axes)
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        Jm = self._out((N, self.n), q2, tm)
        check(lib().rtbhip_jacobm(self._handle(), self._ptr(q2, tm), N, host_ptr(small(tool, 16)), mask,
                                  self._ptr(Jm, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return Jm[0].reshape(self.n, 1) if single else Jm

    _REPRESENTATIONS = {"rpy/xyz": 0, "rpy/zyx": 1, "eul": 2, "exp": 3}

    def jacob0_analytical(self, q, representation="rpy/xyz", tool=None):
        """Analytical Jacobian in the base frame: (6,n), or (N,6,n) for a trajectory (reference ETS.jacob0_analytical
        robot/ETS.py:1562-1626: rotvelxform(R, inverse=True, full=True) @ jacob0)."""
        if representation not in self._REPRESENTATIONS:
            raise ValueError("representation must be one of %s" % ", ".join(self._REPRESENTATIONS))
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        Ja = self._out((N, 6, self.n), q2, tm)
        check(lib().rtbhip_jacob0_analytical(self._handle(), self._ptr(q2, tm), N, host_ptr(small(tool, 16)),
                                             self._REPRESENTATIONS[representation], self._ptr(Ja, tm),
                                             MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return Ja[0] if single else Ja

    def partial_fkine0(self, q, n=3, tool=None):
        """n-th partial derivative of the forward kinematics (robot/ETS.py:1821-2013): n = 1 is jacob0,
        n = 2 hessian0, n >= 3 the (n_joints, ..., 6, n_joints) tensor; a 2-D q adds a leading batch axis."""
        n = int(n)
        if n < 1:
            raise ValueError("n must be >= 1")
        if n == 1:
            return self.jacob0(q, tool=tool)
        if n == 2:
            return self.hessian0(q, tool=tool)
        q2, single, tm = self._shape_q(q)
        N = q2.shape[0]
        out = self._out((N,) + (self.n,) * (n - 1) + (6, self.n), q2, tm)
        check(lib().rtbhip_partial_fkine0(self._handle(), self._ptr(q2, tm), N, host_ptr(small(tool, 16)), n,
                                          self._ptr(out, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return out[0] if single else out

    # ------------------------------------------------------------ inverse kinematics
    def _ik(self, Tep, q0, ilimit, slimit, tol, mask, joint_limits, k, method, flavour, seed, nullspace=None, qp=None):
        n = self.n
        tm = is_torch(Tep) and Tep.is_cuda
        if tm:
            Tq = Tep.reshape(-1, 4, 4).contiguous()
            _lib.note_device(Tq)
            single = Tep.dim() == 2
        else:
            if hasattr(Tep, "A") and not isinstance(Tep, np.ndarray):
                Tep = np.array([x.A for x in Tep]) if len(Tep) > 1 else Tep.A
            a = as_numeric(Tep, "Tep")
            single = a.ndim == 2
            if a.shape[-2:] != (4, 4):
                raise ValueError("Tep must be a 4x4 SE3 matrix")
            Tq = np.ascontiguousarray(a.reshape(-1, 4, 4))
        N = Tq.shape[0]
        if flavour == 1 and single and q0 is not None and not is_torch(q0):
            rows = as_numeric(q0, "q0")
            if rows.ndim == 2 and rows.shape[0] > 1 and rows.shape[1] == n:
                # IKSolver.solve (robot/IK.py:226-240): for ONE pose a (k, n) q0 holds the start vectors of the first k searches; the
                # searches after them start from random vectors.  One search at a time from the supplied rows, then the rest in one call.
                slimit = int(slimit)
                if slimit < 1 or int(ilimit) < 1:
                    raise _lib.RtbHipError("librtbhip error -1: ik_lm: ilimit and slimit must be >= 1")      # what the entry point says
                iters, used = 0, 0
                out = None
                for row in rows[:slimit]:
                    out = self._ik(Tep, row, ilimit, 1, tol, mask, joint_limits, k, method, flavour, seed, nullspace, qp)
                    used += 1
                    iters += int(out[3][0])
                    if int(out[2][0]):
                        break
                if not int(out[2][0]) and used < slimit:
                    out = self._ik(Tep, None, ilimit, slimit - used, tol, mask, joint_limits, k, method, flavour, seed, nullspace, qp)
                    used += int(out[4][0])
                    iters += int(out[3][0])
                _, qo, ok, it, se, E = out
                it = it.copy() if isinstance(it, np.ndarray) else it.clone()
                se = se.copy() if isinstance(se, np.ndarray) else se.clone()
                it[0], se[0] = iters, used
                return True, qo, ok, it, se, E
        q0p = None
        if q0 is not None:
            if tm:
                q0p = q0.reshape(-1, n).contiguous()
                if q0p.shape[0] == 1 and N > 1:
                    q0p = q0p.expand(N, n).contiguous()
            else:
                q0p = as_numeric(q0, "q0").reshape(-1, n)
                if q0p.shape[0] == 1 and N > 1:
                    q0p = np.repeat(q0p, N, axis=0)
                q0p = np.ascontiguousarray(q0p)
            if q0p.shape[0] != N:
                raise ValueError("q0 must be (n,) or (N,n)")
        if isinstance(method, str):
            if method not in _METHODS:
                # the reference dispatches on the first letter (core/fknm.cpp:481-495)
                method = {"s": "sugihara", "w": "wampler"}.get(method[:1], "chan")
            method = _METHODS[method]
        if tm:
            import torch
            qo = torch.empty((N, n), dtype=torch.float64, device=Tq.device)
            ok = torch.empty((N,), dtype=torch.int32, device=Tq.device)
            it = torch.empty((N,), dtype=torch.int32, device=Tq.device)
            se = torch.empty((N,), dtype=torch.int32, device=Tq.device)
            E = torch.empty((N,), dtype=torch.float64, device=Tq.device)
        else:
            qo = np.empty((N, n)); ok = np.empty(N, np.int32); it = np.empty(N, np.int32)
            se = np.empty(N, np.int32); E = np.empty(N)
        we = small(mask, 6)
        kq, km, ps, pi = nullspace if nullspace is not None else (0.0, 0.0, 0.1, None)
        if qp is not None:
            kj, ks = qp
            check(lib().rtbhip_ik_qp(self._handle(), self._ptr(Tq, tm), N, self._ptr(q0p, tm), int(ilimit), int(slimit), float(tol),
                                     int(bool(joint_limits)), host_ptr(we), int(seed) & 0xFFFFFFFFFFFFFFFF, float(kj), float(ks), float(kq),
                                     float(km), float(ps), host_ptr(pi), self._ptr(qo, tm), self._ptr(ok, tm), self._ptr(it, tm),
                                     self._ptr(se, tm), self._ptr(E, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
            return single, qo, ok, it, se, E
        check(lib().rtbhip_ik_lm_nullspace(self._handle(), self._ptr(Tq, tm), N, self._ptr(q0p, tm), int(ilimit), int(slimit),
                                           float(tol), int(bool(joint_limits)), host_ptr(we), float(k), int(method), int(flavour),
                                           int(seed) & 0xFFFFFFFFFFFFFFFF, float(kq), float(km), float(ps), host_ptr(pi),
                                           self._ptr(qo, tm), self._ptr(ok, tm), self._ptr(it, tm),
                                           self._ptr(se, tm), self._ptr(E, tm), MEM_DEVICE if tm else MEM_HOST, self._stream(tm)))
        return single, qo, ok, it, se, E

    def ik_LM(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, k=1.0,
              method="chan", seed=0):
        """Levenberg-Marquardt IK, loop resident on the GPU (reference ETS.ik_LM robot/ETS.py:2014-2170
        -> IK_LM_c core/fknm.cpp:394-525).  One Tep -> the reference's 5-tuple
        (q, success, iterations, searches, residual); Tep (N,4,4) -> the same tuple of arrays."""
        single, q, ok, it, se, E = self._ik(Tep, q0, ilimit, slimit, tol, mask, joint_limits, k, method, 0, seed)
        if single:
            return q[0], int(ok[0]), int(it[0]), int(se[0]), float(E[0])
        return q, ok, it, se, E

    def ik_GN(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, pinv=True,
              pinv_damping=0.0, seed=0):
        """Gauss-Newton IK on the GPU (reference ETS.ik_GN robot/ETS.py:2316-2442 -> IK_GN_c): same search loop,
        minimum-norm step of (J^T W J) dq = J^T W e.  `pinv=False` on a 6-joint arm is the same unique step;
        on a redundant arm the reference's QR branch returns a different (basic) solution of the singular system."""
        single, q, ok, it, se, E = self._ik(Tep, q0, ilimit, slimit, tol, mask, joint_limits, 0.0, "gn", 0, seed)
        if single:
            return q[0], int(ok[0]), int(it[0]), int(se[0]), float(E[0])
        return q, ok, it, se, E

    def ik_NR(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, pinv=True,
              pinv_damping=0.0, seed=0):
        """Newton-Raphson IK on the GPU (reference ETS.ik_NR robot/ETS.py:2172-2298 -> IK_NR_c): dq = J^+ e with
        the damped pseudo-inverse (core/ik.cpp:211-226).  Redundant arms need pinv (the reference forces it)."""
        if not pinv and self.n != 6:
            pinv = True                                       # ik.cpp:128-129
        single, q, ok, it, se, E = self._ik(Tep, q0, ilimit, slimit, tol, mask, joint_limits,
                                            float(pinv_damping) if pinv else 0.0, "nr", 0, seed)
        if single:
            return q[0], int(ok[0]), int(it[0]), int(se[0]), float(E[0])
        return q, ok, it, se, E

    def ikine_LM(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 k=1.0, method="chan", kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        """The Python solver's flavour of LM (reference ETS.ikine_LM robot/ETS.py:2618-2637 ->
        IK_LM.solve/_solve/step robot/IK.py:174-367,994-1017): E is tested after the step and q is
        wrapped with Python's %.  kq / km / ps / pi: the null-space motion of robot/IK.py:507-576 (joint-limit
        avoidance and manipulability maximisation; as in the reference it acts only when kq > 0)."""
        self._no_extra("ikine_LM", kwargs)
        single, q, ok, it, se, E = self._ik(Tep, q0, ilimit, slimit, tol, mask, joint_limits, k, method, 1,
                                            0 if seed is None else seed, self._nullspace(kq, km, ps, pi))
        if is_torch(q):
            q, ok, it, se, E = (x.cpu().numpy() for x in (q, ok, it, se, E))
        if single:
            good = bool(ok[0])
            return IKSolution(q=q[0], success=good, iterations=int(it[0]), searches=int(se[0]),
                              residual=float(E[0]), reason="Success" if good else "iteration and search limit reached")
        allok = bool(ok.all())  # aggregate exactly like reference robot/IK.py:263-290
        return IKSolution(q=q, success=allok, iterations=int(it.sum()), searches=int(se.sum()),
                          residual=float(E.min()) if len(E) else float("inf"),
                          reason="" if allok else "iteration and search limit reached",
                          each={"success": ok.astype(bool), "iterations": it, "searches": se, "residual": E})

    def _nullspace(self, kq, km, ps, pi):
        """(kq, km, ps, pi[n]): the influence distance is a scalar or one value per joint (robot/IK.py:519-520, :1441-1442)."""
        pv = np.asarray(pi.detach().cpu().numpy() if is_torch(pi) else pi, dtype=np.float64).reshape(-1)
        if pv.size == 1:
            pv = np.full(max(1, self.n), float(pv[0]))
        if pv.size != self.n:
            raise ValueError("pi must be a scalar or one influence distance per joint (%d)" % self.n)
        return (float(kq), float(km), float(ps), np.ascontiguousarray(pv))

    def _ikine_pinv(self, name, Tep, q0, ilimit, slimit, tol, mask, joint_limits, seed, pinv, kq, km, ps=0.0, pi=0.3, qp=None):
        if qp is None and not pinv and self.n != 6:
            return self._ikine_no_inverse(Tep, q0, slimit, seed)
        single, q, ok, it, se, E = self._ik(Tep, q0, ilimit, slimit, tol, mask, joint_limits, 0.0, name, 1,
                                            0 if seed is None else seed, self._nullspace(kq, km, ps, pi), qp=qp)
        if is_torch(q):
            q, ok, it, se, E = (x.cpu().numpy() for x in (q, ok, it, se, E))
        if single:
            good = bool(ok[0])
            return IKSolution(q=q[0], success=good, iterations=int(it[0]), searches=int(se[0]),
                              residual=float(E[0]), reason="Success" if good else "iteration and search limit reached")
        allok = bool(ok.all())
        return IKSolution(q=q, success=allok, iterations=int(it.sum()), searches=int(se.sum()),
                          residual=float(E.min()) if len(E) else float("inf"),
                          reason="" if allok else "iteration and search limit reached",
                          each={"success": ok.astype(bool), "iterations": it, "searches": se, "residual": E})

    def _ikine_no_inverse(self, Tep, q0, slimit, seed):
        """ikine_NR / ikine_GN with pinv=False on a chain whose Jacobian is not square: in the reference numpy.linalg.inv raises
        LinAlgError in the FIRST step of every search, which the solver loop catches (robot/IK.py:317-323: "abandon search and try
        again"), so after slimit searches of one counted iteration each it returns a failed IKSolution holding the last start
        vector, residual 0.0 and the LinAlgError count in `reason` (:349-367).  Nothing is computed in the reference and nothing is
        launched here; the start vectors are this backend's restart draws (rtbhip_ik_restart)."""
        tm = is_torch(Tep)
        a = Tep.detach().cpu().numpy() if tm else (Tep.A if hasattr(Tep, "A") and not isinstance(Tep, np.ndarray) else np.asarray(Tep))
        if a.shape[-2:] != (4, 4):
            raise ValueError("Tep must be a 4x4 SE3 matrix")
        single = a.ndim == 2
        N = 1 if single else a.shape[0]
        slimit = int(slimit)
        n = self.n
        q0a = None if q0 is None else as_numeric(q0.detach().cpu().numpy() if is_torch(q0) else q0, "q0").reshape(-1, n)
        q = np.empty((N, n))
        for t in range(N):
            if q0a is not None and slimit == 1:
                q[t] = q0a[min(t, q0a.shape[0] - 1)]
            else:
                q[t] = self.ik_restart(0 if seed is None else seed, t, slimit - 1)
        reason = "iteration and search limit reached, %d numpy.LinAlgError encountered" % slimit
        if single:
            return IKSolution(q=q[0], success=False, iterations=slimit, searches=slimit, residual=0.0, reason=reason)
        z = np.zeros(N)
        return IKSolution(q=q, success=False, iterations=slimit * N, searches=slimit * N, residual=0.0, reason=reason,
                          each={"success": np.zeros(N, bool), "iterations": np.full(N, slimit, np.int32), "searches": np.full(N, slimit, np.int32), "residual": z})

    @staticmethod
    def _no_extra(name, kwargs):
        """The reference forwards **kwargs to the solver class (robot/IK.py: IK_LM.__init__ :860-885, IK_NR :673-697, IK_GN :1113-1137,
        IK_QP :1313-...), which hands them to IKSolver.__init__ (:149) -- and that takes named arguments only: an unknown keyword is
        a TypeError there; it is one here."""
        if kwargs:
            raise TypeError("%s() got an unexpected keyword argument '%s'" % (name, sorted(kwargs)[0]))

    def ikine_NR(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 pinv=False, kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        """The Python Newton-Raphson solver (reference ETS.ikine_NR robot/ETS.py:2639-2776 -> IK_NR robot/IK.py:579-763):
        q += pinv(J) e inside the Python solver's loop semantics (flavour 1)."""
        self._no_extra("ikine_NR", kwargs)
        return self._ikine_pinv("nr", Tep, q0, ilimit, slimit, tol, mask, joint_limits, seed, pinv, kq, km, ps, pi)

    def ikine_GN(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 pinv=False, kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        """The Python Gauss-Newton solver (reference ETS.ikine_GN robot/ETS.py:2778-2915 -> IK_GN robot/IK.py:1020-1220;
        its step is the same pinv(J) e)."""
        self._no_extra("ikine_GN", kwargs)
        return self._ikine_pinv("nr", Tep, q0, ilimit, slimit, tol, mask, joint_limits, seed, pinv, kq, km, ps, pi)

    def ikine_QP(self, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 kj=1.0, ks=1.0, kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        """The quadratic-programming solver (reference ETS.ikine_QP robot/ETS.py:2932-3110 -> IK_QP robot/IK.py:1222-1520;
        note the two defaults of the reference: kj = 1.0 here, 0.01 in the IK_QP class).  Each step minimises
        kj/2 |dq|^2 + ks/(2 sum|e|) |slack|^2 - jacobm.dq/km subject to J dq + slack = e and, when kq > 0, to one velocity-damper
        row per joint inside the influence distance pi of a limit; solved per lane on the device (closed form without rows, a
        primal-dual active set with them: csrc/ik_device.h) -- no qpsolvers dependency.  km > 0 / kq > 0 need 6..12 joints."""
        self._no_extra("ikine_QP", kwargs)
        return self._ikine_pinv("qp", Tep, q0, ilimit, slimit, tol, mask, joint_limits, seed, True, kq, km, ps, pi, qp=(kj, ks))

    def ik_restart(self, seed, target, search):
        """The restart vector the device generator yields (test hook, rtbhip_ik_restart)."""
        out = np.empty(self.n)
        check(lib().rtbhip_ik_restart(self._handle(), int(seed) & 0xFFFFFFFFFFFFFFFF, int(target), int(search),
                                      host_ptr(out)))
        return out

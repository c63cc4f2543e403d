"""oracle/sm_standin.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A stand-in for the slice of spatialmath-python (pyproject.toml:22 `spatialmath-python>=1.1.16`: a third-party dependency of the
reference that is neither vendored nor installable here) that the reference's robot/ET.py, robot/ETS.py and robot/IK.py touch
when their NUMERIC paths -- and the symbolic fall-back of ETS.eval / ETS.jacob0 -- run.  It exists so that those files can be
executed unmodified (oracle/ref_classes.py); it computes nothing of the kinematics, which stay in whichever `fknm` module the
classes are bound to.  Everything below restates the published behaviour of the spatialmath function of the same name for the
argument kinds the reference passes; anything else raises so that a silent divergence is impossible.
"""
import math
import types

import numpy as np

try:
    import sympy
except ImportError:                                  # pragma: no cover
    sympy = None


def issymbol(x):
    """spatialmath.base.symbolic.issymbol: a sympy expression (robot/ET.py:59,84,163,315)."""
    return sympy is not None and isinstance(x, sympy.Expr)


def _sc(theta):
    if issymbol(theta):
        return sympy.sin(theta), sympy.cos(theta), object
    return math.sin(theta), math.cos(theta), np.float64


def _unit(theta, unit):
    if unit.lower().startswith("deg") and not issymbol(theta):
        return math.radians(theta)
    return theta


def trotx(theta, unit="rad"):
    s, c, dt = _sc(_unit(theta, unit))
    return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=dt)


def troty(theta, unit="rad"):
    s, c, dt = _sc(_unit(theta, unit))
    return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1]], dtype=dt)


def trotz(theta, unit="rad"):
    s, c, dt = _sc(_unit(theta, unit))
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=dt)


def getvector(v, dim=None, out="array", dtype=np.float64):
    """spatialmath.base.argcheck.getvector for the uses at robot/ET.py:69,446 and robot/ETS.py (q vectors)."""
    if np.isscalar(v) or issymbol(v):
        v = [v]
    a = np.asarray(v)
    if a.dtype != object:
        a = a.astype(dtype)
    if a.ndim == 2 and 1 in a.shape:
        a = a.reshape(-1)
    if a.ndim != 1:
        raise ValueError("expecting a vector")
    if dim is not None and a.shape[0] != dim:
        raise ValueError("incorrect vector length: expected %d, got %d" % (dim, a.shape[0]))
    if out == "array":
        return a
    if out == "list":
        return list(a)
    if out in ("sequence", "tuple"):
        return tuple(a)
    if out == "row":
        return a.reshape(1, -1)
    if out == "col":
        return a.reshape(-1, 1)
    raise ValueError("invalid output specifier")


def getmatrix(m, shape, dtype=np.float64):
    """spatialmath.base.argcheck.getmatrix as robot/ETS.py:1080 calls it: (None, None) -> a 2-D array, a vector becoming one row."""
    a = np.asarray(m)
    if a.dtype != object:
        a = a.astype(dtype)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        if shape[0] is not None and shape[1] is None:
            a = a.reshape(shape[0], -1)
        elif shape[1] is not None and shape[0] is None:
            a = a.reshape(-1, shape[1])
        else:
            a = a.reshape(1, -1)
    if a.ndim != 2:
        raise ValueError("expecting a matrix")
    for have, want in zip(a.shape, shape):
        if want is not None and have != want:
            raise ValueError("matrix has the wrong shape")
    return a


def verifymatrix(m, shape):
    if not isinstance(m, np.ndarray) or m.shape != tuple(shape):
        raise TypeError("input must be a numPy ndarray of shape %s" % (shape,))


def t2r(T):
    return np.asarray(T)[:3, :3]


def tr2jac(T):
    """spatialmath.base.tr2jac: blkdiag(R, R) (robot/ETS.py, the Python fall-back of jacobe)."""
    R = t2r(T)
    J = np.zeros((6, 6), dtype=np.asarray(T).dtype)
    J[:3, :3] = R
    J[3:, 3:] = R
    return J


def simplify(x):
    return sympy.simplify(x) if issymbol(x) else x


# ---- the further names robot/Link.py, DHLink.py, Dynamics.py, BaseRobot.py, Robot.py and DHRobot.py import (oracle/ref_classes.load_dh)
def isvector(v, dim=None):
    """spatialmath.base.argcheck.isvector: a 1-D (or 1 x n / n x 1) array-like of the given length."""
    try:
        a = np.asarray(v)
    except Exception:
        return False
    if a.ndim == 2 and 1 in a.shape:
        a = a.reshape(-1)
    return a.ndim == 1 and a.dtype != object and (dim is None or a.shape[0] == dim)


def ismatrix(m, shape):
    """spatialmath.base.argcheck.ismatrix: an ndarray of that shape, None standing for any extent."""
    if not isinstance(m, np.ndarray) or m.ndim != 2:
        return False
    return all(want is None or have == want for have, want in zip(m.shape, shape))


def isscalar(x):
    """spatialmath.base.argcheck.isscalar: a real number (int, float, numpy scalar) or a sympy expression."""
    return isinstance(x, (int, float, np.integer, np.floating)) or issymbol(x)


def getunit(v, unit="rad", dim=None):
    """spatialmath.base.argcheck.getunit: the value(s) in radians."""
    a = np.asarray(v, dtype=np.float64) if not np.isscalar(v) else v
    if unit.lower().startswith("deg"):
        return np.radians(a)
    return a


def transl(x, y=None, z=None):
    """spatialmath.base.transl(x, y, z) / transl(v): the 4x4 of a pure translation (models/DH/Panda.py:159)."""
    v = np.asarray(x, dtype=np.float64).reshape(-1) if y is None else np.array([x, y, z], dtype=np.float64)
    T = np.eye(4)
    T[:3, 3] = v
    return T


def islistof(value, what, n=None):
    """spatialmath.base.argcheck.islistof: a list / tuple whose members all are `what` (a type or a predicate), of length n if given."""
    if not isinstance(value, (list, tuple)):
        return False
    if n is not None and len(value) != n:
        return False
    if isinstance(what, type):
        return all(isinstance(x, what) for x in value)
    return all(what(x) for x in value)


def tr2x(T, representation="rpy/xyz"):
    """spatialmath.base.tr2x: [t, Gamma] -- the restatement the oracle already holds (oracle/oracle.py: tr2x)."""
    from . import oracle as _o
    return _o.tr2x(np.asarray(T, dtype=np.float64), representation)


def tr2rpy(T, unit="rad", order="zyx", check=False):
    """spatialmath.base.tr2rpy for the order tools/p_servo.py:97 asks for ("zyx", the default): the restatement the oracle already
    holds for SE3.rpy() (oracle/poe.py: tr2rpy_zyx, singular branch included)."""
    if order not in ("zyx", "vehicle"):
        raise NotImplementedError("spatialmath stand-in: tr2rpy order %r is not restated" % (order,))
    from . import poe as _p
    rpy = _p.tr2rpy_zyx(np.asarray(T, dtype=np.float64)[:3, :3])
    return np.degrees(rpy) if unit == "deg" else rpy


def unitvec_norm(v, tol=20):
    """spatialmath.base.unitvec_norm (spatialmath-python >= 1.1, base/vectors.py): (v / |v|, |v|), or (None, None) for a vector shorter than
    tol * eps.  tools/urdf/urdf.py:1713."""
    v = getvector(v)
    nm = float(np.linalg.norm(v))
    if abs(nm) > tol * np.finfo(np.float64).eps:
        return v / nm, nm
    return None, None


def angvec2r(theta, v, unit="rad", tol=20):
    """spatialmath.base.angvec2r (base/transforms3d.py): Rodrigues' formula  R = I + sin(theta) S + (1 - cos(theta)) S^2,  S = skew(v / |v|);
    the identity for a vector shorter than tol * eps.  tools/urdf/urdf.py:1714."""
    v = getvector(v, 3)
    if np.linalg.norm(v) < tol * np.finfo(np.float64).eps:
        return np.eye(3)
    th = math.radians(theta) if unit == "deg" else float(theta)
    x, y, z = v / np.linalg.norm(v)
    S = np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])
    return np.eye(3) + math.sin(th) * S + (1.0 - math.cos(th)) * (S @ S)


def numjac(f, x, dx=1e-8, SO=0, SE=0):
    """spatialmath.base.numjac: forward-difference Jacobian of f at x.  f returns a vector, or -- SE=3 -- a 4x4 pose, in which case a
    column is [dt / dx ; vex(dR R^T) / dx] (the spatial velocity per unit joint rate), or -- SO=3 -- a rotation matrix."""
    x = np.asarray(x, dtype=np.float64)
    f0 = np.asarray(f(x))
    cols = []
    for i in range(len(x)):
        xi = x.copy()
        xi[i] += dx
        fi = np.asarray(f(xi))
        if SE == 3 or SO == 3:
            R0, Ri = f0[:3, :3], fi[:3, :3]
            S = ((Ri - R0) / dx) @ R0.T
            w = 0.5 * np.array([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]])
            cols.append(np.r_[(fi[:3, 3] - f0[:3, 3]) / dx, w] if SE == 3 else w)
        else:
            cols.append((fi - f0) / dx)
    return np.array(cols).T


def numhess(J, x, dx=1e-8):
    """spatialmath.base.numhess: H[i] = (J(x + dx e_i) - J(x)) / dx."""
    x = np.asarray(x, dtype=np.float64)
    J0 = np.asarray(J(x))
    out = []
    for i in range(len(x)):
        xi = x.copy()
        xi[i] += dx
        out.append((np.asarray(J(xi)) - J0) / dx)
    return np.array(out)


def trnorm(T):
    """spatialmath.base.trnorm: the rotation part made orthonormal again -- n = o x a, o = a x n, columns normalised."""
    T = np.asarray(T, dtype=np.float64)
    o, a = T[:3, 1], T[:3, 2]
    n = np.cross(o, a)
    o = np.cross(a, n)
    R = np.stack([n / np.linalg.norm(n), o / np.linalg.norm(o), a / np.linalg.norm(a)], axis=1)
    out = np.eye(4)
    out[:3, :3] = R
    out[:3, 3] = T[:3, 3]
    return out


def rot2jac(R, representation=None):
    """spatialmath.base.rot2jac: blkdiag(R, R) (robot/Dynamics.py, the operational-space terms)."""
    J = np.zeros((6, 6))
    J[:3, :3] = R
    J[3:, 3:] = R
    return J


def _not_offered(name):
    def f(*a, **k):
        raise NotImplementedError("spatialmath stand-in: %s is not restated (oracle/sm_standin.py)" % name)
    f.__name__ = name
    return f


class SE3:
    """spatialmath.SE3 as far as robot/ET.py, robot/ETS.py and robot/IK.py use it: construction from a 4x4 / a stack (with
    check=False), Empty() / append() (ETS.fkine, robot/ETS.py:1006-1017), isinstance, len, iteration, `.A`, `.inv()`, `.t`, `.R`, `*`."""

    def __init__(self, arg=None, y=None, z=None, check=True):
        if y is not None and z is not None:                   # SE3(x, y, z): a pure translation
            self._data = [transl(float(arg), float(y), float(z))]
        elif arg is None:
            self._data = [np.eye(4)]
        elif isinstance(arg, SE3):
            self._data = [a.copy() for a in arg._data]
        else:
            A = np.asarray(arg)
            if A.dtype != object:
                A = A.astype(np.float64)
            if A.ndim == 1 and A.size == 3:                  # SE3(t): a pure translation (tools/urdf/utils.py:30, urdf.py:1708)
                self._data = [transl(float(A[0]), float(A[1]), float(A[2]))]
            elif A.ndim == 2:
                self._data = [A]
            elif A.ndim == 3:
                self._data = list(A)
            else:
                raise ValueError("bad argument to SE3 constructor")
        for a in self._data:
            if a.shape != (4, 4):
                raise ValueError("SE3 needs 4x4 matrices")

    @classmethod
    def Empty(cls):
        x = cls()
        x._data = []
        return x

    def append(self, other):
        self._data.extend(other._data)

    @classmethod
    def Alloc(cls, n):                                        # robot/Robot.py:669 (fkine_all fills a preallocated stack)
        x = cls()
        x._data = [np.eye(4) for _ in range(n)]
        return x

    def __setitem__(self, i, v):
        self._data[i] = (v.A if isinstance(v, SE3) else np.asarray(v, dtype=np.float64)).copy()

    def copy(self): return SE3(self)                          # robot/DHRobot.py:1058 (fkine_all starts from a copy of the base)
    def __len__(self): return len(self._data)
    def __iter__(self): return (SE3(a, check=False) for a in self._data)
    def __getitem__(self, i): return SE3(self._data[i], check=False)

    @property
    def A(self): return self._data[0] if len(self._data) == 1 else np.array(self._data)

    @property
    def t(self): return self.A[..., :3, 3]

    @property
    def R(self): return self.A[..., :3, :3]

    def inv(self):
        out = []
        for a in self._data:
            X = np.eye(4, dtype=a.dtype)
            X[:3, :3] = a[:3, :3].T
            X[:3, 3] = -a[:3, :3].T @ a[:3, 3]
            out.append(X)
        return SE3(np.array(out) if len(out) > 1 else out[0], check=False)

    def __mul__(self, other):
        if isinstance(other, SE3):
            return SE3(self.A @ other.A, check=False)
        if isinstance(other, np.ndarray) and other.ndim == 2 and other.shape[0] == 3 and len(self._data) == 1:
            # spatialmath BasePoseMatrix.__mul__: a pose times a (3, N) array transforms its COLUMNS as points, h2e(A @ e2h(P)).  This is what
            # tools/urdf/urdf.py:1716 `SE3.RPY(joint.rpy) * R` (R a 3x3 ndarray) evaluates to: RPY's rotation times R (its translation is zero)
            A = self._data[0]
            return A[:3, :3] @ other + A[:3, 3:4]
        return NotImplemented

    def __imul__(self, other):
        return self.__mul__(other)

    # the constructors robot/DHRobot.py, DHLink.py and the models/DH files use
    @classmethod
    def Rx(cls, theta, unit="rad"): return cls(trotx(theta, unit), check=False)

    @classmethod
    def Ry(cls, theta, unit="rad"): return cls(troty(theta, unit), check=False)

    @classmethod
    def Rz(cls, theta, unit="rad"): return cls(trotz(theta, unit), check=False)

    @classmethod
    def RPY(cls, roll, pitch=None, yaw=None, order="zyx", unit="rad"):
        """spatialmath SE3.RPY, zyx order: R = Rz(yaw) Ry(pitch) Rx(roll)."""
        if pitch is None:
            roll, pitch, yaw = roll
        if order != "zyx":
            raise NotImplementedError("spatialmath stand-in: SE3.RPY order %r is not restated" % order)
        return cls(trotz(yaw, unit) @ troty(pitch, unit) @ trotx(roll, unit), check=False)

    @classmethod
    def Trans(cls, x, y=None, z=None):
        v = np.asarray(x, dtype=np.float64).reshape(-1) if y is None else np.array([x, y, z], dtype=np.float64)
        T = np.eye(4)
        T[:3, 3] = v
        return cls(T, check=False)

    @classmethod
    def Tx(cls, x): return cls.Trans(x, 0.0, 0.0)

    @classmethod
    def Ty(cls, y): return cls.Trans(0.0, y, 0.0)

    @classmethod
    def Tz(cls, z): return cls.Trans(0.0, 0.0, z)

    def __eq__(self, other):
        return isinstance(other, SE3) and len(self) == len(other) and all(np.array_equal(a, b) for a, b in zip(self._data, other._data))

    __hash__ = None


class SE2:
    """Only ever an isinstance target on the paths exercised (the 2-D classes ET2 / ETS2 are not run)."""

    def __init__(self, *a, **k):
        raise NotImplementedError("spatialmath stand-in: SE2 is not restated")


def modules():
    """(spatialmath, spatialmath.base) module objects to be placed in sys.modules while the reference files are loaded."""
    sm = types.ModuleType("spatialmath")
    smb = types.ModuleType("spatialmath.base")
    for f in (trotx, troty, trotz, issymbol, getvector, getmatrix, verifymatrix, t2r, tr2jac, simplify, isvector, ismatrix, getunit, rot2jac, transl, islistof, tr2x, tr2rpy, numjac, numhess, trnorm,
              unitvec_norm, angvec2r):
        setattr(smb, f.__name__, f)
    for name in ("tr2eul", "trlog", "trot2", "transl2", "tr2xyt", "tr2jac2", "rotvelxform", "r2x", "rotvelxform_inv_dot"):
        setattr(smb, name, _not_offered(name))
    argcheck = types.ModuleType("spatialmath.base.argcheck")
    for f in (getvector, getmatrix, verifymatrix, isscalar, isvector, ismatrix, getunit):
        setattr(argcheck, f.__name__, f)
    symbolic = types.ModuleType("spatialmath.base.symbolic")
    symbolic.issymbol, symbolic.simplify = issymbol, simplify
    # spatialmath.base.symbolic.sin / cos / sqrt: the sympy function for a symbol, math's otherwise (robot/DHRobot.py:1620, DHLink.py)
    symbolic.sin = lambda x: sympy.sin(x) if issymbol(x) else math.sin(x)
    symbolic.cos = lambda x: sympy.cos(x) if issymbol(x) else math.cos(x)
    symbolic.sqrt = lambda x: sympy.sqrt(x) if issymbol(x) else math.sqrt(x)
    symbolic.pi = (lambda: sympy.S.Pi) if sympy is not None else _not_offered("pi")
    symbolic.zero = (lambda: sympy.S.Zero) if sympy is not None else _not_offered("zero")
    symbolic.symbol = (lambda *a, **k: sympy.symbols(*a, **k)) if sympy is not None else _not_offered("symbol")
    smb.argcheck, smb.symbolic = argcheck, symbolic
    for name in ("Twist3", "SpatialAcceleration", "SpatialVelocity", "SpatialInertia", "SpatialForce"):   # import targets only
        setattr(sm, name, type(name, (), {"__init__": lambda self, *a, **k: (_ for _ in ()).throw(
            NotImplementedError("spatialmath stand-in: this class is not restated"))}))
    # used only by tools/p_servo.py's pure-Python fall-back, which never runs (Angle_Axis does not raise)
    smb.iszerovec = lambda v, tol=20: bool(np.linalg.norm(v) < tol * np.finfo(np.float64).eps)
    smb.norm = lambda v: float(np.linalg.norm(v))
    smb.isscalar = isscalar
    sm.SE3, sm.SE2, sm.base = SE3, SE2, smb
    return sm, smb

"""rtbhip.compat.fknm -- the function table of the reference's `fknm` extension module (core/fknm.cpp:23-93) over
librtbhip.so.  Argument tuples, return shapes / memory orders and the `TypeError("Symbolic value")` control-flow signal
(core/fknm.cpp:1304-1318) are the reference's; what differs is stated per function.

    ET_init, ET_update, ETS_init            handles (plain Python objects instead of PyCapsules)
    ETS_fkine, ETS_jacob0, ETS_jacobe       one configuration as the reference; a 2-D q of N > 1 rows is a batch
    ETS_hessian0, ETS_hessiane              from q or -- the reference's usual call -- from the supplied Jacobian
    IK_LM_c, IK_GN_c, IK_NR_c               one Tep as the reference; Tep (N,4,4) solves N targets in one launch
    Angle_Axis                              one pair as the reference; stacks of poses are a batch
    ET_T, Robot_link_T                      (scene-graph refresh) routed through the same kernels
r2q (core/fknm.cpp:1283-1302) is not imported by any reference Python module and is not offered.

Restarts of the IK searches come from the device's counter-based generator keyed by (seed, target, search, joint); the
reference uses an unseeded std::rand (core/ik.cpp:293).  `set_ik_seed` changes the seed (default 0).
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import lib, check, host_ptr, as_numeric, rtbhip_et, MEM_HOST, MEM_DEVICE, is_torch

_IK_SEED = [0]


def set_ik_seed(seed):
    _IK_SEED[0] = int(seed) & 0xFFFFFFFFFFFFFFFF


def _numeric(x):
    """_check_array_type (core/fknm.cpp:1304-1318): anything that is not a number array is `TypeError("Symbolic value")`."""
    if hasattr(x, "A") and not isinstance(x, np.ndarray) and not is_torch(x):
        x = x.A
    return as_numeric(x)


class _ET:
    """What ET_init's capsule holds (struct ET, core/structs.h:42-56).  As in the reference the constant matrix is
    BORROWED: the array object is kept and re-read whenever a chain is (re)built."""
    __slots__ = ("isstaticsym", "isjoint", "isflip", "jindex", "axis", "T", "qlim", "version")

    def __init__(self, isstaticsym, isjoint, isflip, jindex, jointtype, T, qlim):
        self.version = 0
        self.set(isstaticsym, isjoint, isflip, jindex, jointtype, T, qlim)

    def set(self, isstaticsym, isjoint, isflip, jindex, jointtype, T, qlim):
        if not isinstance(T, np.ndarray) or not isinstance(qlim, np.ndarray):
            raise TypeError("T and qlim must be numpy arrays")            # the "O!" of PyArg_ParseTuple
        self.isstaticsym, self.isjoint, self.isflip = int(isstaticsym), int(isjoint), int(isflip)
        self.jindex, self.axis = int(jindex), int(jointtype)
        self.T, self.qlim = T, np.array([float(qlim.flat[0]), float(qlim.flat[1])])
        self.version += 1


class _ETS:
    """What ETS_init's capsule holds (struct ETS, core/structs.h:25-40): the ET list; the device chain is compiled on
    first use and again whenever an ET_update touched one of its elements."""

    def __init__(self, ets, n, m):
        ets = list(ets)
        if len(ets) < m or any(not isinstance(e, _ET) for e in ets[:m]):
            raise TypeError("ETS_init needs a list of ET handles")
        self.ets, self.n, self.m = ets[:m], int(n), int(m)
        self._handle, self._versions = None, None

    def handle(self):
        versions = tuple(e.version for e in self.ets)
        if self._handle is not None and versions == self._versions:
            return self._handle
        self._drop()
        arr = (rtbhip_et * max(1, self.m))()
        lows, highs = [], []
        for i, e in enumerate(self.ets):
            if e.isstaticsym:
                raise TypeError("Symbolic value")
            arr[i].kind = e.axis if e.isjoint else _lib.ET_CONST
            arr[i].flip, arr[i].jindex = e.isflip, e.jindex
            flat = np.asarray(e.T, dtype=np.float64).reshape(4, 4) if not e.isjoint else np.eye(4)
            for r in range(4):
                for c in range(4):
                    arr[i].T[4 * r + c] = flat[r, c]                     # logical (row, col): memory order does not matter
            if e.isjoint:
                lows.append(e.qlim[0]); highs.append(e.qlim[1])
        ql = np.ascontiguousarray(lows + highs, dtype=np.float64) if lows else None
        h = C.c_uint64(0)
        check(lib().rtbhip_chain_create(arr, self.m, host_ptr(ql), C.byref(h)))
        self._handle, self._versions = h.value, versions
        n, qw = C.c_int32(0), C.c_int32(0)
        check(lib().rtbhip_chain_info(self._handle, C.byref(n), None, C.byref(qw)))
        self.q_width = qw.value
        return self._handle

    def _drop(self):
        if self._handle is not None:
            try:
                lib().rtbhip_chain_destroy(self._handle)
            except Exception:
                pass
            self._handle = None

    def __del__(self):
        self._drop()


def ET_init(isstaticsym, isjoint, isflip, jindex, jointtype, T, qlim):
    """core/fknm.cpp:1182-1239."""
    return _ET(isstaticsym, isjoint, isflip, jindex, jointtype, T, qlim)


def ET_update(et, isstaticsym, isjoint, isflip, jindex, jointtype, T, qlim):
    """core/fknm.cpp:1116-1180 (returns a handle to the SAME element, as the reference returns a new capsule on it)."""
    if not isinstance(et, _ET):
        raise TypeError("ET_update needs an ET handle")
    et.set(isstaticsym, isjoint, isflip, jindex, jointtype, T, qlim)
    return et


def ETS_init(ets, n, m):
    """core/fknm.cpp:1066-1114."""
    return _ETS(ets, n, m)


def _ets(obj):
    if not isinstance(obj, _ETS):
        raise ValueError("PyCapsule_GetPointer called with incorrect name")   # what a wrong capsule raises in the reference
    return obj


def _q_rows(e, q):
    """(rows (N, q_width), single): 1-D, (1,n) and (n,1) are one configuration (core/fknm.cpp:964-988).
    One deliberate difference: for a ONE-joint chain the reference's rule makes every (N,1) array "a single q vector" (it then evaluates
    the first element only, :973-979); here (N,1) with N > 1 on a one-joint chain is the trajectory of N configurations it looks like
    ((1,1) is still one configuration).  Documented because the module otherwise follows the reference's shape rules to the letter."""
    a = _numeric(q)
    e.handle()
    if a.ndim == 0:
        a = a.reshape(1)
    single = a.ndim == 1 or a.shape[0] == 1 or (a.shape[1] == 1 and e.q_width != 1)
    if a.ndim > 2:
        raise ValueError("q must be 1-D or 2-D")
    a = a.reshape(1, -1) if single else a
    if a.shape[1] < e.q_width:
        raise ValueError("q has %d columns, the chain needs %d" % (a.shape[1], e.q_width))
    return np.ascontiguousarray(a[:, :e.q_width]), single


def _se3(x):
    if x is None:
        return None
    a = _numeric(x)
    if a.shape != (4, 4):
        raise ValueError("expected a 4x4 matrix")
    return np.ascontiguousarray(a)            # LOGICAL 4x4, row-major for the ABI whatever order the caller used


def ETS_fkine(ets, q, base, tool, include_base):
    """core/fknm.cpp:923-1064: one q -> (4,4) Fortran-order; a trajectory (N,n) -> (N,4,4) C-order."""
    e = _ets(ets)
    rows, single = _q_rows(e, q)
    b = None
    if base is not None:
        _numeric(base)                         # type-checked whether or not it is used (fknm.cpp:1012-1024)
        if include_base:
            b = _se3(base)
    t = _se3(tool)
    N = rows.shape[0]
    T = np.empty((N, 4, 4))
    check(lib().rtbhip_fkine(e.handle(), host_ptr(rows), N, host_ptr(b), host_ptr(t), host_ptr(T), MEM_HOST, None))
    return np.asfortranarray(T[0]) if single else T


def _jac(ets, q, tool, frame):
    e = _ets(ets)
    rows, single = _q_rows(e, q)
    N = rows.shape[0]
    J = np.empty((N, 6, e.n))
    check(lib().rtbhip_jacob(e.handle(), host_ptr(rows), N, host_ptr(_se3(tool)), frame, host_ptr(J), MEM_HOST, None))
    return np.asfortranarray(J[0]) if single else J


def ETS_jacob0(ets, q, tool):
    """core/fknm.cpp:785-850: (6,n) Fortran-order; a 2-D q of N > 1 rows -> (N,6,n)."""
    return _jac(ets, q, tool, 0)


def ETS_jacobe(ets, q, tool):
    """core/fknm.cpp:852-921."""
    return _jac(ets, q, tool, 1)


def _hess(ets, q, J, tool, frame):
    e = _ets(ets)
    if J is not None:
        a = _numeric(J)                        # logical (6,n) [or (N,6,n)], whatever its memory order
        single = a.ndim == 2
        a3 = np.ascontiguousarray(a.reshape((1,) + a.shape) if single else a)
        if a3.shape[1:] != (6, e.n):
            raise ValueError("J must be (6,%d)" % e.n)
        H = np.empty((a3.shape[0], e.n, 6, e.n))
        check(lib().rtbhip_hessian_from_jacobian(host_ptr(a3), a3.shape[0], e.n, host_ptr(H), MEM_HOST, None))
        return H[0] if single else H
    rows, single = _q_rows(e, q)
    N = rows.shape[0]
    H = np.empty((N, e.n, 6, e.n))
    check(lib().rtbhip_hessian(e.handle(), host_ptr(rows), N, host_ptr(_se3(tool)), frame, host_ptr(H), MEM_HOST, None))
    return H[0] if single else H


def ETS_hessian0(ets, q, J, tool):
    """core/fknm.cpp:583-682: (n,6,n) C-order, from J when it is given (q and tool are then not looked at), else from q."""
    return _hess(ets, q, J, tool, 0)


def ETS_hessiane(ets, q, J, tool):
    """core/fknm.cpp:684-783."""
    return _hess(ets, q, J, tool, 1)


def _ik(ets, Tep, q0, ilimit, slimit, tol, reject_jl, we, lam, method):
    e = _ets(ets)
    T = _numeric(Tep)
    single = T.ndim == 2
    if T.shape[-2:] != (4, 4):
        raise ValueError("Tep must be 4x4")
    T3 = np.ascontiguousarray(T.reshape(-1, 4, 4))
    N = T3.shape[0]
    q0a = None
    if q0 is not None:
        q0a = _numeric(q0).reshape(-1, e.n)
        if q0a.shape[0] == 1 and N > 1:
            q0a = np.repeat(q0a, N, axis=0)
        q0a = np.ascontiguousarray(q0a)
        if q0a.shape[0] != N:
            raise ValueError("q0 must be (n,) or (N,n)")
    wea = None if we is None else np.ascontiguousarray(_numeric(we).reshape(6))
    q = np.empty((N, e.n)); ok = np.empty(N, np.int32); it = np.empty(N, np.int32); se = np.empty(N, np.int32); E = np.empty(N)
    check(lib().rtbhip_ik_lm(e.handle(), host_ptr(T3), N, host_ptr(q0a), int(ilimit), int(slimit), float(tol), int(bool(reject_jl)),
                             host_ptr(wea), float(lam), int(method), 0, _IK_SEED[0], host_ptr(q), host_ptr(ok), host_ptr(it),
                             host_ptr(se), host_ptr(E), MEM_HOST, None))
    if single:
        return q[0], int(ok[0]), int(it[0]), int(se[0]), float(E[0])
    return q, ok, it, se, E


def IK_LM_c(ets, Tep, q0, ilimit, slimit, tol, reject_jl, we, lam, method):
    """core/fknm.cpp:394-525 -> (q, solution, iterations, searches, E); the method is chosen by the FIRST LETTER of the
    string, anything but 's' / 'w' is Chan (fknm.cpp:481-495)."""
    if not isinstance(method, str):
        raise TypeError("argument 10 must be str")
    m = {"s": 2, "w": 1}.get(method[:1], 0)
    return _ik(ets, Tep, q0, ilimit, slimit, tol, reject_jl, we, lam, m)


def IK_GN_c(ets, Tep, q0, ilimit, slimit, tol, reject_jl, we, use_pinv, pinv_damping):
    """core/fknm.cpp:279-392 (Gauss-Newton; the minimum-norm step serves both of the reference's branches, see ik_device.h)."""
    return _ik(ets, Tep, q0, ilimit, slimit, tol, reject_jl, we, 0.0, 3)


def IK_NR_c(ets, Tep, q0, ilimit, slimit, tol, reject_jl, we, use_pinv, pinv_damping):
    """core/fknm.cpp:164-277 (Newton-Raphson with the damped pseudo-inverse; a redundant arm forces pinv, ik.cpp:128-129)."""
    e = _ets(ets)
    if not use_pinv and e.n != 6:
        use_pinv = 1
    return _ik(ets, Tep, q0, ilimit, slimit, tol, reject_jl, we, float(pinv_damping) if use_pinv else 0.0, 4)


def Angle_Axis(Te, Tep):
    """core/fknm.cpp:112-162: (6,) for one pair; stacks of poses give (N,6)."""
    A, B = _numeric(Te), _numeric(Tep)
    if A.shape[-2:] != (4, 4) or B.shape[-2:] != (4, 4):
        raise ValueError("poses must be 4x4")
    single = A.ndim == 2 and B.ndim == 2
    A3, B3 = np.ascontiguousarray(A.reshape(-1, 4, 4)), np.ascontiguousarray(B.reshape(-1, 4, 4))
    N = max(A3.shape[0], B3.shape[0])
    e = np.empty((N, 6))
    check(lib().rtbhip_angle_axis(host_ptr(A3), A3.shape[0], host_ptr(B3), B3.shape[0], host_ptr(e), MEM_HOST, None))
    return e[0] if single else e


def ET_T(et, eta):
    """core/fknm.cpp:1241-1281: the 4x4 (Fortran order) of one elementary transform at joint value eta -- for a joint the
    one-element chain through rtbhip_fkine, for a constant a copy of its matrix.  A non-float eta is `TypeError("Symbolic value")` exactly as in the reference."""
    if not isinstance(et, _ET):
        raise ValueError("PyCapsule_GetPointer called with incorrect name")
    if et.isstaticsym:
        raise TypeError("Symbolic value")
    val = 0.0
    if eta is not None:
        if not isinstance(eta, float):
            raise TypeError("Symbolic value")
        val = eta
    if not et.isjoint:
        # a constant element: the reference copies the stored matrix (_ET_T, core/methods.cpp:354-370) -- nothing to compute
        return np.asfortranarray(np.array(et.T, dtype=np.float64).reshape(4, 4))
    one = _ET(0, et.isjoint, et.isflip, 0, et.axis, et.T, et.qlim)
    chain = _ETS([one], 1 if et.isjoint else 0, 1)
    T = np.empty((1, 4, 4))
    qrow = np.array([[val]])
    check(lib().rtbhip_fkine(chain.handle(), host_ptr(qrow), 1, None, None, host_ptr(T), MEM_HOST, None))
    return np.asfortranarray(T[0])


def Robot_link_T(ets_list, T_list, self_q, q):
    """core/fknm.cpp:526-581: refresh every link's transform array in place from its ETS (scene-graph update).  Each T is
    written through its own strides, so Fortran-order arrays -- what the reference allocates -- are filled correctly."""
    if not isinstance(self_q, np.ndarray):
        raise TypeError("argument 3 must be numpy.ndarray")
    try:
        qv = _numeric(q) if q is not None else None
    except TypeError:
        qv = None                                   # the reference falls back to self_q on a failed type check
    if qv is None:
        qv = np.asarray(self_q, dtype=np.float64)
    qv = np.ascontiguousarray(qv.reshape(-1))
    for e, T in zip(ets_list, T_list):
        e = _ets(e)
        e.handle()
        row = np.zeros((1, max(1, e.q_width)))
        k = min(e.q_width, qv.size)
        row[0, :k] = qv[:k]
        out = np.empty((1, 4, 4))
        check(lib().rtbhip_fkine(e.handle(), host_ptr(row), 1, None, None, host_ptr(out), MEM_HOST, None))
        T[...] = out[0]
    return None

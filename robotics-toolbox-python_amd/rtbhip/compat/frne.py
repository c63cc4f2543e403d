"""rtbhip.compat.frne -- the function table of the reference's `frne` extension module (core/frne.c:42-62) over
librtbhip.so: `init(njoints, mdh, L, gravity)`, `frne(robot, q, qd, qdd, gravity, fext)`, `delete(robot)`, with the
reference's argument meaning (L = the flattened (n,24) block of DHRobot._init_rne, robot/DHRobot.py:1342-1358; gravity as
DHRobot hands it over, i.e. already negated, :1360-1361, :1449).  One (q, qd, qdd) -> a list of n floats as the reference
returns; 2-D arrays of N rows -> an (N,n) ndarray from one launch."""
import ctypes as C

import numpy as np

from .._lib import lib, check, host_ptr, as_numeric, MEM_HOST


class _Robot:
    def __init__(self, n, handle, gravity):
        self.n, self.handle, self.gravity = n, handle, gravity


def init(njoints, mdh, L, gravity):
    """core/frne.c:233-299 (the gravity stored here is overwritten by every frne call, as in the reference, frne.c:141-155)."""
    n = int(njoints)
    Lf = np.ascontiguousarray(np.asarray(list(L) if not isinstance(L, np.ndarray) else L, dtype=np.float64).reshape(-1))
    if Lf.size < 24 * n:
        raise ValueError("L must hold 24 values per link")
    g = np.asarray(list(gravity), dtype=np.float64).reshape(-1)[:3].copy()
    h = C.c_uint64(0)
    check(lib().rtbhip_dyn_create(host_ptr(Lf), n, int(mdh), C.byref(h)))
    return _Robot(n, h.value, g)


def _robot(r):
    if not isinstance(r, _Robot) or r.handle is None:
        raise ValueError("PyCapsule_GetPointer called with incorrect name")
    return r


def frne(robot, q, qd, qdd, gravity, fext):
    """core/frne.c:106-230."""
    r = _robot(robot)
    g = np.asarray(list(gravity) if not isinstance(gravity, np.ndarray) else gravity, dtype=np.float64).reshape(-1)
    if g.size < 3:
        raise ValueError("gravity vector too short")
    r.gravity = np.ascontiguousarray(g[:3])
    a = [as_numeric(x) for x in (q, qd, qdd)]
    single = a[0].ndim <= 1
    rows = []
    for x, name in zip(a, ("q", "qd", "qdd")):
        x2 = x.reshape(1, -1) if single else x.reshape(x.shape[0], -1)
        if x2.shape[1] < r.n:
            raise ValueError("%s iterator exhausted at element %d" % (name, x2.shape[1]))
        rows.append(np.ascontiguousarray(x2[:, :r.n]))
    N = rows[0].shape[0]
    if rows[1].shape[0] != N or rows[2].shape[0] != N:
        raise ValueError("q, qd and qdd must have the same number of rows")
    f = np.asarray(list(fext) if not isinstance(fext, np.ndarray) else fext, dtype=np.float64).reshape(-1)
    if f.size < 6:
        raise ValueError("fext iterator exhausted at element %d" % f.size)
    f = np.ascontiguousarray(f[:6])
    tau = np.empty((N, r.n))
    check(lib().rtbhip_rne(r.handle, host_ptr(rows[0]), host_ptr(rows[1]), host_ptr(rows[2]), N, host_ptr(r.gravity), host_ptr(f),
                           host_ptr(tau), MEM_HOST, None))
    return [float(v) for v in tau[0]] if single else tau


def delete(robot):
    """core/frne.c:80-103."""
    r = _robot(robot)
    check(lib().rtbhip_dyn_destroy(r.handle))
    r.handle = None
    return 1

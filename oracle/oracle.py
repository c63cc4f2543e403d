"""oracle/oracle.py -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

ctypes front-end of oracle/liboracle.so (the plain-C restatement, rtb_oracle.c).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status: pinned (see rtb_oracle.h header).
"""
import ctypes as C
import os
import subprocess
import math
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class _CChain(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("kind", _ip), ("flip", _ip), ("jindex", _ip),
                ("consts", _dp)]


def build():
    subprocess.check_call(["make", "-s", "-f", os.path.join(_HERE, "Makefile"), "lib"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        for name in ("oracle_fkine", "oracle_jacob", "oracle_hessian", "oracle_angle_axis",
                     "oracle_ik_lm", "oracle_ikine_lm", "oracle_rne_dh", "oracle_dh_A",
                     "oracle_dh_fkine"):
            getattr(_LIB, name).restype = None
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return a if shape is None else np.ascontiguousarray(a.reshape(shape))


def _cchain(ch):
    keep = (np.ascontiguousarray(ch.kind, dtype=np.int32), np.ascontiguousarray(ch.flip, dtype=np.int32),
            np.ascontiguousarray(ch.jindex, dtype=np.int32), _f64(ch.consts, (-1, 16)))
    cc = _CChain(int(ch.m), int(ch.n), keep[0].ctypes.data_as(_ip), keep[1].ctypes.data_as(_ip),
                 keep[2].ctypes.data_as(_ip), keep[3].ctypes.data_as(_dp))
    return cc, keep


def fkine(ch, q, base=None, tool=None):
    q = _f64(q)
    q2 = q.reshape(-1, q.shape[-1]) if q.ndim > 1 else q.reshape(1, -1)
    N, w = q2.shape
    T = np.empty((N, 4, 4))
    cc, keep = _cchain(ch)
    b, t = _f64(base, (4, 4)), _f64(tool, (4, 4))
    lib().oracle_fkine(C.byref(cc), _d(q2), C.c_long(N), C.c_int(w), _d(b), _d(t), _d(T))
    return T


def jacob(ch, q, tool=None, frame=0):
    q = _f64(q)
    q2 = q.reshape(-1, q.shape[-1]) if q.ndim > 1 else q.reshape(1, -1)
    N, w = q2.shape
    J = np.empty((N, 6, ch.n))
    cc, keep = _cchain(ch)
    t = _f64(tool, (4, 4))
    lib().oracle_jacob(C.byref(cc), _d(q2), C.c_long(N), C.c_int(w), _d(t), C.c_int(frame), _d(J))
    return J


def jacob0(ch, q, tool=None):
    return jacob(ch, q, tool, 0)


def jacobe(ch, q, tool=None):
    return jacob(ch, q, tool, 1)


def hessian0(ch, q, tool=None):
    J = jacob(ch, q, tool, 0)
    H = np.empty((J.shape[0], ch.n, 6, ch.n))
    for i in range(J.shape[0]):
        Ji = np.ascontiguousarray(J[i])
        lib().oracle_hessian(C.c_int(ch.n), _d(Ji), _d(H[i]))
    return H


def hessian(ch, q, tool=None, frame=0):
    J = jacob(ch, q, tool, frame)
    H = np.empty((J.shape[0], ch.n, 6, ch.n))
    for i in range(J.shape[0]):
        Ji = np.ascontiguousarray(J[i])
        lib().oracle_hessian(C.c_int(ch.n), _d(Ji), _d(H[i]))
    return H


def jacob_dot(ch, q, qd, tool=None, frame=0):
    """Robot.jacob0_dot (robot/Robot.py:1063-1098, representation=None): tensordot(H, qd, (0, 0))."""
    H = hessian(ch, q, tool, frame)
    qd = _f64(qd).reshape(H.shape[0], ch.n)
    return np.array([np.tensordot(H[i], qd[i], (0, 0)) for i in range(H.shape[0])])


def tr2x(T, representation="rpy/xyz"):
    """spatialmath-python 1.1.x `tr2x` (third-party, absent from the reference tree): [t, Gamma] with Gamma = tr2rpy(order
    "xyz" / "zyx") = (roll, pitch, yaw), tr2eul = (phi, theta, psi) or the rotation vector of trlog.  Conventions:
    R = Rx(yaw) Ry(pitch) Rz(roll) for "xyz", Rz(yaw) Ry(pitch) Rx(roll) for "zyx", Rz(phi) Ry(theta) Rz(psi) for "eul"."""
    R, t = np.asarray(T)[:3, :3], np.asarray(T)[:3, 3]
    if representation == "rpy/xyz":
        g = [-math.atan2(R[0, 1], R[0, 0]), math.atan2(R[0, 2], math.hypot(R[1, 2], R[2, 2])), -math.atan2(R[1, 2], R[2, 2])]
    elif representation == "rpy/zyx":
        g = [math.atan2(R[2, 1], R[2, 2]), -math.atan2(R[2, 0], math.hypot(R[0, 0], R[1, 0])), math.atan2(R[1, 0], R[0, 0])]
    elif representation == "eul":
        phi = math.atan2(R[1, 2], R[0, 2])
        sp, cp = math.sin(phi), math.cos(phi)
        g = [phi, math.atan2(cp * R[0, 2] + sp * R[1, 2], R[2, 2]), math.atan2(-sp * R[0, 0] + cp * R[1, 0], -sp * R[0, 1] + cp * R[1, 1])]
    elif representation == "exp":
        l = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        nrm = np.linalg.norm(l)
        th = math.atan2(nrm, np.trace(R) - 1)
        g = list(l * (th / nrm)) if nrm > 1e-12 else [0.0, 0.0, 0.0]
    else:
        raise ValueError(representation)
    return np.r_[t, g]


def _rot(axis, a):
    c, s = math.cos(a), math.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def rotvelxform_inverse(R, representation):
    """What `rotvelxform(R, inverse=True, full=True, representation=...)` returns (ETS.py:1624-1625): blkdiag(I, A^-1), A
    mapping the rates of Gamma to the angular velocity.  Built from the ANGLES of tr2x (the device builds it from the entries
    of R): omega = a x1' + R_a b x2' + R_a R_b c x3' for R = R_a(x1) R_b(x2) R_c(x3); left Jacobian of SO(3) for "exp"."""
    g = tr2x(np.r_[np.c_[R, np.zeros(3)], [[0, 0, 0, 1]]], representation)[3:]
    ex, ey, ez = np.eye(3)
    if representation == "rpy/xyz":
        A = np.c_[_rot("x", g[2]) @ _rot("y", g[1]) @ ez, _rot("x", g[2]) @ ey, ex]
    elif representation == "rpy/zyx":
        A = np.c_[_rot("z", g[2]) @ _rot("y", g[1]) @ ex, _rot("z", g[2]) @ ey, ez]
    elif representation == "eul":
        A = np.c_[ez, _rot("z", g[0]) @ ey, _rot("z", g[0]) @ _rot("y", g[1]) @ ez]
    else:
        th = np.linalg.norm(g)
        S = np.array([[0, -g[2], g[1]], [g[2], 0, -g[0]], [-g[1], g[0], 0]])
        A = np.eye(3) if th < 1e-9 else np.eye(3) + (1 - math.cos(th)) / th ** 2 * S + (th - math.sin(th)) / th ** 3 * S @ S
    X = np.eye(6)
    X[3:, 3:] = np.linalg.inv(A)
    return X


def jacob0_analytical(ch, q, representation="rpy/xyz", tool=None):
    """ETS.jacob0_analytical (robot/ETS.py:1622-1626): A @ jacob0(q)."""
    T, J = fkine(ch, q, tool=tool), jacob(ch, q, tool, 0)
    return np.array([rotvelxform_inverse(T[i][:3, :3], representation) @ J[i] for i in range(J.shape[0])])


def jacob0_dot_analytical(ch, q, qd, representation="rpy/xyz", tool=None, dx=1e-8):
    """Robot.jacob0_dot with a `representation` (robot/Robot.py:1090-1098): H = numhess(jacob0_analytical, q) -- spatialmath's
    forward difference H[i] = (J(q + dx e_i) - J(q)) / dx with dx = 1e-8 (third-party, restated) -- then tensordot(H, qd, (0, 0))."""
    q = _f64(q).reshape(-1, ch.n)
    qd = _f64(qd).reshape(-1, ch.n)
    out = np.zeros((q.shape[0], 6, ch.n))
    I = np.eye(ch.n)
    for s in range(q.shape[0]):
        J0 = jacob0_analytical(ch, q[s], representation, tool)[0]
        H = np.stack([(jacob0_analytical(ch, q[s] + I[:, i] * dx, representation, tool)[0] - J0) / dx for i in range(ch.n)], axis=0)
        out[s] = np.tensordot(H, qd[s], (0, 0))
    return out


def link_frames(ch, q, marks, base=None):
    """DHRobot.fkine_all / Robot.fkine_all (robot/DHRobot.py:1058-1064, robot/Robot.py:667-698): Tj = base; Tj *= A_k(q) and
    every partial product kept.  Frame m = base * (first marks[m] elementary transforms), each built as the reference's
    rx/ry/rz/tx/ty/tz do (core/fknm.cpp:1320-1555; flip negates the coordinate).  (N, nmarks, 4, 4)."""
    from . import chains as _ch
    q = _f64(q).reshape(-1, max(1, int(ch.jindex.max()) + 1 if ch.n else 1)) if ch.n else np.zeros((np.atleast_2d(q).shape[0], 0))
    names = ["Rx", "Ry", "Rz", "tx", "ty", "tz"]
    out = np.empty((q.shape[0], len(marks), 4, 4))
    for i in range(q.shape[0]):
        T = np.eye(4) if base is None else np.array(base, dtype=float)
        prefix = [T.copy()]
        for k in range(ch.m):
            if ch.kind[k] == 6:
                A = ch.consts[k].reshape(4, 4)
            else:
                eta = q[i, ch.jindex[k]]
                A = _ch.elementary(names[ch.kind[k]], -eta if ch.flip[k] else eta)
            T = T @ A
            prefix.append(T.copy())
        for m, k in enumerate(marks):
            out[i, m] = prefix[k]
    return out


def partial_fkine0(ch, q, n, tool=None):
    """ETS.partial_fkine0 (robot/ETS.py:1862-2013) for one configuration: dT[c][..., l, k, :, j] from the
    product rule on H[k, :, j] = J_w[:, k] x J[:, j].  Same term bookkeeping as the reference
    (`add_indices` / `add_pdi`, :1888-1927), plain loops."""
    q = _f64(q).reshape(-1)
    nj = ch.n
    dT = [jacob(ch, q, tool, 0)[0], hessian(ch, q, tool, 0)[0]]
    terms = [([1], [0])]                       # positions in the digit vector (j, k, l, ...)
    while len(dT) < n:
        c = len(dT) + 1                        # order of the tensor being built
        nxt = []
        for a, b in terms:
            nxt.append((a + [c - 1], b))
            nxt.append((a, b + [c - 1]))
        terms = nxt
        pd = np.zeros([nj] * (c - 1) + [6, nj])
        for digits in np.ndindex(*([nj] * c)):                 # digits[0] = j, digits[1] = k, ...
            trn = np.zeros(3)
            rot = np.zeros(3)
            for a, b in terms:
                ta, tb = dT[len(a) - 1], dT[len(b) - 1]
                ia = tuple(digits[i] for i in reversed(a[1:]))
                ib = tuple(digits[i] for i in reversed(b[1:]))
                wa = ta[ia + (slice(3, 6), digits[a[0]])]
                rot += np.cross(wa, tb[ib + (slice(3, 6), digits[b[0]])])
                trn += np.cross(wa, tb[ib + (slice(0, 3), digits[b[0]])])
            lead = tuple(reversed(digits[1:]))
            pd[lead + (slice(0, 3), digits[0])] = trn
            pd[lead + (slice(3, 6), digits[0])] = rot
        dT.append(pd)
    return dT[n - 1]


def _axes_list(axes):
    if isinstance(axes, str):
        if axes.startswith("all"):
            return [True] * 6
        if axes.startswith("trans"):
            return [True] * 3 + [False] * 3
        if axes.startswith("rot"):
            return [False] * 3 + [True] * 3
        raise ValueError("axes must be all, trans or rot")
    return [bool(a) for a in axes]


def manipulability(ch, q, axes="all", tool=None, method="yoshikawa"):
    """ETS.manipulability (robot/ETS.py:1766-1819): yoshikawa / invcondition / minsingular."""
    ax = _axes_list(axes)
    J = jacob(ch, q, tool, 0)
    out = np.zeros(J.shape[0])
    for k in range(J.shape[0]):
        Jk = J[k][ax, :]
        if method == "invcondition":
            out[k] = 1 / np.linalg.cond(Jk)
        elif method == "minsingular":
            out[k] = np.linalg.svd(Jk, compute_uv=False)[-1]
        elif Jk.shape[0] == Jk.shape[1]:
            out[k] = abs(np.linalg.det(Jk))
        else:
            out[k] = np.sqrt(abs(np.linalg.det(Jk @ Jk.T)))
    return out


def jacobm(ch, q, axes="all", tool=None):
    """Robot.jacobm (robot/Robot.py:1185-1235) / ETS.jacobm (robot/ETS.py:1669-1685): (N, n)."""
    ax = _axes_list(axes)
    J = jacob(ch, q, tool, 0)
    H = hessian(ch, q, tool, 0)
    m = manipulability(ch, q, axes, tool)
    out = np.zeros((J.shape[0], ch.n))
    for k in range(J.shape[0]):
        Jk = J[k][ax, :]
        Hk = H[k][:, ax, :]
        b = np.linalg.inv(Jk @ Jk.T)
        for i in range(ch.n):
            c = Jk @ Hk[i].T
            out[k, i] = m[k] * (c.flatten("F")).T @ b.flatten("F")
    return out


def angle_axis(Te, Tep):
    Te, Tep = _f64(Te, (4, 4)), _f64(Tep, (4, 4))
    e = np.empty(6)
    lib().oracle_angle_axis(_d(Te), _d(Tep), _d(e))
    return e


METHODS = {"chan": 0, "wampler": 1, "sugihara": 2}


def ik_lm(ch, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, reject_jl=True, we=None, k=1.0,
          method="chan", restarts=None):
    """C-flavoured LM (reference ETS.ik_LM -> IK_LM_c).  restarts: (slimit+1, n) start vectors."""
    n = ch.n
    if restarts is None:
        raise ValueError("oracle ik_lm needs the restart vectors (the RNG is external)")
    restarts = _f64(restarts, (-1, n))
    assert restarts.shape[0] >= slimit + 1
    cc, keep = _cchain(ch)
    q = np.zeros(n)
    sol, it, search = C.c_int(0), C.c_int(0), C.c_int(0)
    E = C.c_double(0)
    Tep = _f64(Tep, (4, 4))
    q0 = _f64(q0, (n,)) if q0 is not None else None
    we = _f64(we, (6,)) if we is not None else None
    qlim = _f64(ch.qlim, (2, n))
    lib().oracle_ik_lm(C.byref(cc), _d(qlim), _d(Tep), _d(q0), C.c_int(ilimit), C.c_int(slimit),
                       C.c_double(tol), C.c_int(int(bool(reject_jl))), _d(we), C.c_double(k),
                       C.c_int(METHODS[method] if isinstance(method, str) else int(method)),
                       _d(restarts), _d(q), C.byref(sol), C.byref(it), C.byref(search), C.byref(E))
    return q, sol.value, it.value, search.value, E.value


def ikine_lm(ch, Tep, q0s, ilimit=30, slimit=100, tol=1e-6, joint_limits=True, we=None, k=1.0,
             method="chan"):
    """Python-flavoured LM (reference ETS.ikine_LM -> IK.py).  q0s: (slimit, n) start vectors."""
    n = ch.n
    q0s = _f64(q0s, (-1, n))
    assert q0s.shape[0] >= slimit
    cc, keep = _cchain(ch)
    q = np.zeros(n)
    ok, it, se = C.c_int(0), C.c_int(0), C.c_int(0)
    E = C.c_double(0)
    Tep = _f64(Tep, (4, 4))
    we = _f64(we, (6,)) if we is not None else None
    qlim = _f64(ch.qlim, (2, n))
    lib().oracle_ikine_lm(C.byref(cc), _d(qlim), _d(Tep), _d(q0s), C.c_int(ilimit), C.c_int(slimit),
                          C.c_double(tol), C.c_int(int(bool(joint_limits))), _d(we), C.c_double(k),
                          C.c_int(METHODS[method] if isinstance(method, str) else int(method)),
                          _d(q), C.byref(ok), C.byref(it), C.byref(se), C.byref(E))
    return q, ok.value, it.value, se.value, E.value


def rne_dh(L24, mdh, q, qd, qdd, grav_c, fext=None):
    """grav_c is what frne.frne is handed (already negated by DHRobot.rne, DHRobot.py:1449)."""
    L = _f64(L24, (-1, 24))
    n = L.shape[0]
    q, qd, qdd = _f64(q, (-1, n)), _f64(qd, (-1, n)), _f64(qdd, (-1, n))
    N = q.shape[0]
    tau = np.empty((N, n))
    g = _f64(grav_c, (3,))
    f = _f64(fext, (6,)) if fext is not None else None
    lib().oracle_rne_dh(_d(L), C.c_int(n), C.c_int(int(mdh)), _d(q), _d(qd), _d(qdd), C.c_long(N),
                        _d(g), _d(f), _d(tau))
    return tau


def null_sigma(ch, q, ps, pi):
    """_null_Sigma (robot/IK.py:507-539): joint-limit avoidance gradient, (n,1)."""
    pi = pi * np.ones(ch.n) if np.ndim(pi) == 0 else np.asarray(pi, dtype=float)
    S = np.zeros((ch.n, 1))
    for i in range(ch.n):
        qi, ql0, ql1 = q[i], ch.qlim[0, i], ch.qlim[1, i]
        if qi - ql0 <= pi[i]:
            S[i, 0] = -np.power(((qi - ql0) - pi[i]), 2) / np.power((ps - pi[i]), 2)
        if ql1 - qi <= pi[i]:
            S[i, 0] = np.power(((ql1 - qi) - pi[i]), 2) / np.power((ps - pi[i]), 2)
    return -S


def calc_qnull(ch, q, J, kq, km, ps, pi):
    """_calc_qnull (robot/IK.py:542-576), including its guard `kq > 0 or kq > 0` on the projection."""
    grad = np.zeros(ch.n)
    qnull = np.zeros(ch.n)
    if kq > 0:
        grad += (1.0 / kq * null_sigma(ch, q, ps, pi)).flatten()
    if km > 0:
        grad += (1.0 / km * jacobm(ch, q)[0].reshape(ch.n, 1)).flatten()
    if kq > 0 or kq > 0:
        qnull = (np.eye(ch.n) - np.linalg.pinv(J) @ J) @ grad
    return qnull.flatten()


def qp_step(ch, q, e, J, kj, ks, km, kq=0.0, ps=0.0, pi=0.3):
    """IK_QP.step (robot/IK.py:1437-1500), matrix for matrix: Q, c, Aeq, beq, Ain, bin as the reference builds them; the QP itself
    goes to oracle/qp.py (the reference: qpsolvers / quadprog, an absent optional dependency)."""
    from . import qp
    n = ch.n
    pi = pi * np.ones(n) if np.ndim(pi) == 0 else np.asarray(pi, dtype=float)
    Q = np.eye(n + 6)
    Q[:n, :n] *= kj
    Q[n:, n:] = ks * (1 / np.sum(np.abs(e))) * np.eye(6)
    Aeq = np.concatenate((J, np.eye(6)), axis=1)
    beq = e.reshape((6,))
    if kq > 0.0:
        Ain = np.zeros((n + 6, n + 6))
        bin_ = np.zeros(n + 6)
        Ain_l = np.zeros((n, n))
        Bin_l = np.zeros(n)
        for i in range(n):
            ql0, ql1 = ch.qlim[0, i], ch.qlim[1, i]
            if ql1 - q[i] <= pi[i]:
                Bin_l[i] = ((ql1 - q[i]) - ps) / (pi[i] - ps)
                Ain_l[i, i] = 1
            if q[i] - ql0 <= pi[i]:
                Bin_l[i] = -(((ql0 - q[i]) + ps) / (pi[i] - ps))
                Ain_l[i, i] = -1
        Ain[:n, :n] = Ain_l
        bin_[:n] = (1.0 / kq) * Bin_l
    else:
        Ain, bin_ = None, None
    if km > 0.0:
        Jm = jacobm(ch, q)[0].reshape((n,))
        c = np.concatenate(((1.0 / km) * -Jm, np.zeros(6)))
    else:
        c = np.zeros(n + 6)
    xd = qp.solve_qp(Q, c, Ain, bin_, Aeq, beq)
    if xd is None:
        raise np.linalg.LinAlgError("QP Unsolvable")
    return xd[:n]


def ikine_py(ch, Tep, q0s, step="nr", ilimit=30, slimit=100, tol=1e-6, joint_limits=True, we=None, k=1.0,
             kq=0.0, km=0.0, ps=0.0, pi=0.3, method="chan", kj=1.0, ks=1.0):
    """NumPy restatement of the Python solvers' loop, IKSolver._solve (robot/IK.py:297-367), with the steps of
    IK_NR / IK_GN (q += pinv(J) e + qnull, :736-763, :1176-1220), IK_LM (:994-1017) or IK_QP (step "qp", :1437-1500).
    q0s: (slimit, n) starts."""
    n = ch.n
    q0s = _f64(q0s, (-1, n))
    Tep = _f64(Tep, (4, 4))
    We = np.diag(np.ones(6) if we is None else _f64(we, (6,)))
    total_i, E, q = 0, 0.0, q0s[0].copy()
    for search in range(slimit):
        q = q0s[search].copy()
        i = 0
        while i < ilimit:
            i += 1
            Te = fkine(ch, q)[0]
            e = angle_axis(Te, Tep)
            E = 0.5 * e @ We @ e
            J = jacob0(ch, q)[0]
            try:
                if step == "qp":
                    if not np.all(np.isfinite(J)):
                        raise np.linalg.LinAlgError("QP Unsolvable")
                    q = q + qp_step(ch, q, e, J, kj, ks, km, kq, ps, pi)
                    qnull = None
                else:
                    qnull = calc_qnull(ch, q, J, kq, km, ps, pi)
                if step == "qp":
                    pass
                elif step in ("nr", "gn"):
                    q = q + np.linalg.pinv(J) @ e + qnull
                else:
                    g = J.T @ We @ e
                    Wn = {"chan": k * E, "wampler": k, "sugihara": E + k}[method] * np.eye(n)
                    q = q + np.linalg.inv(J.T @ We @ J + Wn) @ g + qnull
            except np.linalg.LinAlgError:                  # IK.py:320-323: abandon the search
                break
            if E < tol:
                q = (q + np.pi) % (2 * np.pi) - np.pi
                ok = bool(np.all(q >= ch.qlim[0]) and np.all(q <= ch.qlim[1]))
                if not ok and joint_limits:
                    break
                return q, 1, total_i + i, search + 1, E
        total_i += i
    return q, 0, total_i, slimit, E


def nofriction_L24(L24):
    """Dynamics.nofriction(True, True) on the 24-double link block: B = 0, Tc = 0 (robot/Dynamics.py:146-183)."""
    L = _f64(L24, (-1, 24)).copy()
    L[:, 21:24] = 0.0
    return L


def inertia_dh(L24, mdh, q):
    """Dynamics.inertia (robot/Dynamics.py:748-763): per configuration, rne over n rows with qdd = I."""
    L = _f64(L24, (-1, 24)); n = L.shape[0]
    q = _f64(q, (-1, n))
    out = np.zeros((q.shape[0], n, n))
    for k, qk in enumerate(q):
        out[k] = rne_dh(L, mdh, np.tile(qk, (n, 1)), np.zeros((n, n)), np.eye(n), [0, 0, 0])
    return out


def coriolis_dh(L24, mdh, q, qd):
    """Dynamics.coriolis (robot/Dynamics.py:811-861), statement for statement."""
    L = nofriction_L24(L24); n = L.shape[0]
    q, qd = _f64(q, (-1, n)), _f64(qd, (-1, n))
    Cm = np.zeros((q.shape[0], n, n))
    Csq = np.zeros((q.shape[0], n, n))
    z = np.zeros(n)
    for k, qk in enumerate(q):
        for i in range(n):
            QD = np.zeros(n); QD[i] = 1
            Csq[k, :, i] = Csq[k, :, i] + rne_dh(L, mdh, qk, QD, z, [0, 0, 0])[0]
    for k, (qk, qdk) in enumerate(zip(q, qd)):
        for i in range(n):
            for j in range(i + 1, n):
                QD = np.zeros(n); QD[i] = 1; QD[j] = 1
                tau = rne_dh(L, mdh, qk, QD, z, [0, 0, 0])[0]
                Cm[k, :, j] = Cm[k, :, j] + (tau - Csq[k, :, j] - Csq[k, :, i]) * qdk[i] / 2
                Cm[k, :, i] = Cm[k, :, i] + (tau - Csq[k, :, j] - Csq[k, :, i]) * qdk[j] / 2
        Cm[k] = Cm[k] + Csq[k] @ np.diag(qdk)
    return Cm


def accel_dh(L24, mdh, q, qd, torque, grav_c):
    """Dynamics.accel (robot/Dynamics.py:484-505): M from n unit accelerations, tau_0, numpy.linalg.solve."""
    L = _f64(L24, (-1, 24)); n = L.shape[0]
    q, qd, torque = _f64(q, (-1, n)), _f64(qd, (-1, n)), _f64(torque, (-1, n))
    out = np.zeros_like(q)
    for k in range(q.shape[0]):
        M = rne_dh(L, mdh, np.tile(q[k], (n, 1)), np.zeros((n, n)), np.eye(n), [0, 0, 0])
        tau = rne_dh(L, mdh, q[k], qd[k], np.zeros(n), grav_c)[0]
        out[k] = np.linalg.solve(M, torque[k] - tau)
    return out


def dh_fkine(dh7, mdh, q, base=None, tool=None):
    dh = _f64(dh7, (-1, 7))
    n = dh.shape[0]
    q = _f64(q, (-1, n))
    N = q.shape[0]
    T = np.empty((N, 4, 4))
    b, t = _f64(base, (4, 4)), _f64(tool, (4, 4))
    lib().oracle_dh_fkine(_d(dh), C.c_int(n), C.c_int(int(mdh)), _d(q), C.c_long(N), _d(b), _d(t), _d(T))
    return T

"""oracle/erobot.py -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

NumPy restatement of the reference's ETS-robot inverse dynamics, ``Robot.rne``
(robot/Robot.py:1704-1903), statement for statement, with spatialmath's 6-D spatial vectors written
out as explicit 6-vectors / 6x6 matrices ([linear; angular] order, spatialmath/spatialvector.py):

    SE3 * SpatialVelocity/Acceleration   ->  Ad(T) @ v            (motion transform)
    SE3 * SpatialForce                   ->  Ad(T).T @ f          (force transform: the dual of the motion transform)
    SpatialVelocity @ SpatialVelocity    ->  crm(v) @ v'          (spatial motion cross product)
    SpatialVelocity @ SpatialForce/Momentum -> crf(v) @ f = -crm(v).T @ f
    SpatialInertia(m, r)                 ->  [[m 1, -m r^], [m r^, -m r^ r^]]   (no inertia tensor: Robot.py:1797)

spatialmath is a dependency of the reference that is absent from /root/reference and from this image
(pyproject.toml:22 `spatialmath-python>=1.1.16`); the algebra above is its published definition.
Pinned on the closed-form two-link values of the reference's own tests (tests/test_ERobot.py:101-274)
in tests/test_erobot_rne.py.  Parity beyond those cases is unpinned.

A robot is a list of links in the order the reference would hold them (BaseRobot._sort_links: depth
first): dict(name, parent (name or None), ets (oracle.chains item list; a trailing joint item makes the
link a joint), m, r).
"""
import numpy as np

from . import chains


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def Ad(T):
    """Adjoint of an SE3 acting on [linear; angular] motion vectors."""
    R, p = T[:3, :3], T[:3, 3]
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[:3, 3:] = skew(p) @ R
    A[3:, 3:] = R
    return A


def crm(v):
    out = np.zeros((6, 6))
    out[:3, :3] = skew(v[3:])
    out[:3, 3:] = skew(v[:3])
    out[3:, 3:] = skew(v[3:])
    return out


def crf(v):
    return -crm(v).T


def spatial_inertia(m, r):
    C = skew(np.asarray(r, dtype=np.float64))
    I = np.zeros((6, 6))
    I[:3, :3] = m * np.eye(3)
    I[:3, 3:] = -m * C          # m C^T
    I[3:, :3] = m * C
    I[3:, 3:] = -m * C @ C      # m C C^T (zero inertia tensor about the centre of mass)
    return I


def _link_parts(link):
    """(constant 4x4, joint item or None) of one link's ets."""
    T = np.eye(4)
    joint = None
    for item in link["ets"]:
        if isinstance(item, np.ndarray):
            T = T @ item
        elif len(item) > 1 and item[1] is not None:
            T = T @ chains.elementary(item[0], item[1])
        else:
            joint = item
    return T, joint


def _joint_T(item, q):
    axis = item[0]
    flip = bool(item[2]) if len(item) > 2 else False
    return chains.elementary(axis, -q if flip else q)        # ET.A(q): flip negates q (robot/ET.py:313-316)


def _joint_s(item):
    s = np.zeros(6)                                          # robot/ET.py:592-608 (flip is ignored)
    k = chains.AXES[item[0]]
    s[(k - 3) if k >= 3 else (3 + k)] = 1.0
    return s


def erobot_rne(links, q, qd, qdd, gravity=(0, 0, -9.81)):
    names = [l["name"] for l in links]
    parts = [_link_parts(l) for l in links]
    isjoint = [p[1] is not None for p in parts]
    # jindex in link order (auto numbering by the depth-first order the caller supplies) -- or the links' own "jindex" entries when the robot was
    # numbered by hand (BaseRobot.py: "all links must have a jindex, or none have a jindex"); q, qd, qdd are read by jindex (Robot.py:1830-1853),
    # the torques are returned in GROUP order (Q[k, j], :1880)
    jindex, k = [], 0
    for l, j in zip(links, isjoint):
        jindex.append((l.get("jindex", k) if j else None))
        k += 1 if j else 0
    n = k
    q = np.asarray(q, dtype=np.float64).reshape(-1, n)
    qd = np.asarray(qd, dtype=np.float64).reshape(-1, n)
    qdd = np.asarray(qdd, dtype=np.float64).reshape(-1, n)
    # Robot.py:1777-1789
    link_groups, cur = [], []
    for i in range(len(links)):
        cur.append(i)
        if isjoint[i]:
            link_groups.append(cur)
            cur = []
    # Robot.py:1791-1802
    I, s = [], []
    for group in link_groups:
        I_int = np.zeros((6, 6))
        for idx in group:
            I_int = I_int + spatial_inertia(links[idx]["m"], links[idx]["r"])
            if isjoint[idx]:
                s.append(_joint_s(parts[idx][1]))
        I.append(I_int)
    a_grav = -np.r_[np.asarray(gravity, dtype=np.float64), 0, 0, 0]          # Robot.py:1804-1807
    Q = np.zeros((q.shape[0], n))
    for kk in range(q.shape[0]):
        v = [None] * n; a = [None] * n; f = [None] * n; Xup = [None] * n
        for j, group in enumerate(link_groups):                              # forward recursion :1822-1872
            joint = group[-1]
            ji = jindex[joint]
            vJ = s[j] * qd[kk, ji]
            X = np.eye(4)
            for idx in group:
                Tc, item = parts[idx]
                X = X @ (Tc @ _joint_T(item, q[kk, jindex[idx]]) if item is not None else Tc)
            Xup[j] = np.linalg.inv(X)
            first = links[group[0]]
            if first["parent"] is None:
                v[j] = vJ
                a[j] = Ad(Xup[j]) @ a_grav + s[j] * qdd[kk, ji]
            else:
                pidx = names.index(first["parent"])
                gidx = [i for i, g in enumerate(link_groups) if pidx in g][0]
                v[j] = Ad(Xup[j]) @ v[gidx] + vJ
                a[j] = Ad(Xup[j]) @ a[gidx] + s[j] * qdd[kk, ji] + crm(v[j]) @ vJ
            f[j] = I[j] @ a[j] + crf(v[j]) @ (I[j] @ v[j])
        for j in reversed(range(n)):                                         # backward recursion :1875-1893
            group = link_groups[j]
            Q[kk, j] = float(np.sum(f[j] * s[j]))
            first = links[group[0]]
            if first["parent"] is not None:
                pidx = names.index(first["parent"])
                gidx = [i for i, g in enumerate(link_groups) if pidx in g][0]
                f[gidx] = f[gidx] + Ad(Xup[j]).T @ f[j]                      # SE3 * SpatialForce = Ad(T)^T f
    return Q


# ---- the Dynamics-mixin terms of an ETS robot: restatement of robot/Dynamics.py over erobot_rne (TEST INFRASTRUCTURE, as the rest of
# this file).  Pinned in tests/test_erobot_dynamics.py on the reference's OWN DynamicsMixin methods, executed unmodified
# (oracle/ref_classes.load_dh) on a stand-in robot whose `rne` is erobot_rne: the loops below and the reference's loops must agree bit
# for bit, since they call the same function with the same arguments in the same order.
def erobot_inertia(links, q, gravity=(0, 0, -9.81)):
    """Dynamics.inertia (robot/Dynamics.py:744-763): row i of M = rne(q, 0, e_i) without gravity."""
    q = np.atleast_2d(np.asarray(q, dtype=np.float64))
    n = q.shape[1]
    out = np.zeros((q.shape[0], n, n))
    for k, qk in enumerate(q):
        out[k] = erobot_rne(links, np.tile(qk, (n, 1)), np.zeros((n, n)), np.eye(n), (0, 0, 0))
    return out


def erobot_coriolis(links, q, qd):
    """Dynamics.coriolis (robot/Dynamics.py:811-861), statement for statement."""
    q, qd = np.atleast_2d(np.asarray(q, dtype=np.float64)), np.atleast_2d(np.asarray(qd, dtype=np.float64))
    n = q.shape[1]
    Cm, Csq = np.zeros((q.shape[0], n, n)), np.zeros((q.shape[0], n, n))
    z = np.zeros(n)
    for k, qk in enumerate(q):
        for i in range(n):
            QD = np.zeros(n)
            QD[i] = 1
            Csq[k, :, i] = Csq[k, :, i] + erobot_rne(links, qk, QD, z, (0, 0, 0))[0]
    for k, (qk, qdk) in enumerate(zip(q, qd)):
        for i in range(n):
            for j in range(i + 1, n):
                QD = np.zeros(n)
                QD[i] = 1
                QD[j] = 1
                tau = erobot_rne(links, qk, QD, z, (0, 0, 0))[0]
                Cm[k, :, j] = Cm[k, :, j] + (tau - Csq[k, :, j] - Csq[k, :, i]) * qdk[i] / 2
                Cm[k, :, i] = Cm[k, :, i] + (tau - Csq[k, :, j] - Csq[k, :, i]) * qdk[j] / 2
        Cm[k] = Cm[k] + Csq[k] @ np.diag(qdk)
    return Cm


def erobot_accel(links, q, qd, torque, gravity=(0, 0, -9.81)):
    """Dynamics.accel (robot/Dynamics.py:483-505): solve(M, torque - rne(q, qd, 0))."""
    q, qd, torque = (np.atleast_2d(np.asarray(x, dtype=np.float64)) for x in (q, qd, torque))
    n = q.shape[1]
    out = np.zeros((q.shape[0], n))
    for k, (qk, qdk, tk) in enumerate(zip(q, qd, torque)):
        M = erobot_rne(links, np.tile(qk, (n, 1)), np.zeros((n, n)), np.eye(n), (0, 0, 0))
        tau = erobot_rne(links, qk, qdk, np.zeros(n), gravity)[0]
        out[k] = np.linalg.solve(M, tk - tau)
    return out

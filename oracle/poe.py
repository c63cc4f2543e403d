"""oracle/poe.py -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE): product-of-exponentials robots.

A NumPy restatement of the reference's robot/PoERobot.py, function by function:
  * the closed-form PoE kinematics -- PoERobot.fkine (PoERobot.py:209-228), jacob0 (:230-250), jacobe (:252-270);
  * the reference's OWN lowering of a PoE robot to an ETS -- PoELink._ets_world (:42-121) and PoERobot._update_ets
    (:272-318): world frames from the twists, partial transforms between consecutive frames, each written as
    tx ty tz Rz(yaw) Ry(pitch) Rx(roll) with the near-zero elements dropped (np.isclose, :112, :314) and the joint appended.

The arithmetic under those functions is third-party: spatialmath-python (pyproject.toml:22 pins `spatialmath-python>=1.1.16`,
not vendored, not installable here).  What is restated below is its published algorithm for exactly the calls PoERobot.py makes:
Twist3.UnitRevolute / UnitPrismatic, Twist3.exp (-> base.trexp on a 6-vector), SE3.Ad, SE3.inv, SE3.OA (-> base.oa2r),
SE3.rpy (-> base.tr2rpy, order "zyx") and base.skew.

PARITY PIN (how this file is checked, tests/test_04_poe.py): the reference's own test, tests/test_PoERobot.py:14-74, states that
for its two robots (a 2R-P-R arm and a 3R-P arm with arbitrary axes) the closed form and the lowered ETS agree on fkine,
jacob0 and jacobe.  Here the closed form below is compared with the lowered ETS evaluated by the reference's COMPILED fknm
(oracle/_ref) -- two formulations, one of them the reference's own binary -- at the test's own q.  That pins the twist
exponential, the adjoint, the OA frame and the roll-pitch-yaw round trip against each other and against fknm; what it cannot pin
is the free choice inside `oa2r` (which x axis a frame gets), which changes the ET list `ets()` prints but not one number of
fkine / jacob0 / jacobe.
"""
import math

import numpy as np

from . import chains

_EPS = np.finfo(np.float64).eps


# ------------------------------------------------------------------------------------------------ spatialmath pieces
def skew(v):
    """spatialmath.base.skew (used at PoERobot.py:249)."""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def unit_revolute(axis, point, pitch=0.0):
    """Twist3.UnitRevolute(a, q, pitch) as PoERevolute builds it (PoERobot.py:139): w = a/|a|, v = -w x q + pitch w; S = (v, w)."""
    w = np.asarray(axis, dtype=np.float64)
    w = w / np.linalg.norm(w)
    v = -np.cross(w, np.asarray(point, dtype=np.float64)) + pitch * w
    return np.r_[v, w]


def unit_prismatic(axis):
    """Twist3.UnitPrismatic(a) as PoEPrismatic builds it (PoERobot.py:154): w = 0, v = a/|a|."""
    v = np.asarray(axis, dtype=np.float64)
    return np.r_[v / np.linalg.norm(v), np.zeros(3)]


def rodrigues(w, theta):
    """spatialmath.base.rodrigues for a unit w: I + sin(theta) [w] + (1 - cos(theta)) [w]^2."""
    K = skew(w)
    return np.eye(3) + math.sin(theta) * K + (1.0 - math.cos(theta)) * K @ K


def twist_exp(S, theta):
    """Twist3.exp(theta) (PoERobot.py:221-224, :243, :265): base.trexp(S * theta) on a 6-vector -- the vector is split into a unit
    twist and its magnitude (unittwist_norm: |w| when w != 0, else |v|), then R = rodrigues(w, th),
    t = (I th + (1 - cos th)[w] + (th - sin th)[w]^2) v."""
    tw = np.asarray(S, dtype=np.float64) * theta
    T = np.eye(4)
    v, w = tw[:3], tw[3:]
    if np.linalg.norm(tw) < 10 * _EPS:
        return T
    if np.linalg.norm(w) < 10 * _EPS:
        th = np.linalg.norm(v)
    else:
        th = np.linalg.norm(w)
    v, w = v / th, w / th
    K = skew(w)
    T[:3, :3] = rodrigues(w, th)
    T[:3, 3] = (np.eye(3) * th + (1.0 - math.cos(th)) * K + (th - math.sin(th)) * K @ K) @ v
    return T


def Ad(T):
    """SE3.Ad() (PoERobot.py:242, :264, :270): [[R, [t] R], [0, R]] acting on (v, w)."""
    R, t = T[:3, :3], T[:3, 3]
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[:3, 3:] = skew(t) @ R
    A[3:, 3:] = R
    return A


def se3_inv(T):
    """SE3.inv()."""
    X = np.eye(4)
    X[:3, :3] = T[:3, :3].T
    X[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return X


def oa2r(o, a):
    """spatialmath.base.oa2r behind SE3.OA (PoERobot.py:97): n = o x a, o = a x n, columns unit(n), unit(o), unit(a)."""
    o, a = np.asarray(o, dtype=np.float64), np.asarray(a, dtype=np.float64)
    n = np.cross(o, a)
    o = np.cross(a, n)
    unit = lambda x: x / np.linalg.norm(x)
    return np.stack((unit(n), unit(o), unit(a)), axis=1)


def tr2rpy_zyx(R):
    """spatialmath.base.tr2rpy(order="zyx"), the default behind SE3.rpy() (PoERobot.py:100, :302): (roll, pitch, yaw) with
    R = Rz(yaw) Ry(pitch) Rx(roll); at |R[2,0]| = 1 roll is set to 0."""
    rpy = np.zeros(3)
    if abs(abs(R[2, 0]) - 1) < 10 * _EPS:
        rpy[0] = 0.0
        if R[2, 0] < 0:
            rpy[2] = -math.atan2(R[0, 1], R[0, 2])
        else:
            rpy[2] = math.atan2(-R[0, 1], -R[0, 2])
        rpy[1] = -math.asin(min(1.0, max(-1.0, R[2, 0])))
    else:
        rpy[0] = math.atan2(R[2, 1], R[2, 2])
        rpy[2] = math.atan2(R[1, 0], R[0, 0])
        k = int(np.argmax(np.abs([R[0, 0], R[1, 0], R[2, 1], R[2, 2]])))
        if k == 0:
            rpy[1] = -math.atan(R[2, 0] * math.cos(rpy[2]) / R[0, 0])
        elif k == 1:
            rpy[1] = -math.atan(R[2, 0] * math.sin(rpy[2]) / R[1, 0])
        elif k == 2:
            rpy[1] = -math.atan(R[2, 0] * math.sin(rpy[0]) / R[2, 1])
        else:
            rpy[1] = -math.atan(R[2, 0] * math.cos(rpy[0]) / R[2, 2])
    return rpy


# ------------------------------------------------------------------------------------------------ closed form
class PoE:
    """twists: (n,6) rows (v, w); T0: 4x4 -- what PoERobot holds as links[1..n].S and self.T0."""

    def __init__(self, twists, T0):
        self.S = np.asarray(twists, dtype=np.float64).reshape(-1, 6)
        self.T0 = np.asarray(T0, dtype=np.float64).reshape(4, 4)
        self.n = self.S.shape[0]

    def fkine(self, q):
        """PoERobot.fkine (PoERobot.py:209-228): exp(S_1 q_1) ... exp(S_n q_n) T0; rows of a 2-D q one by one."""
        q = np.asarray(q, dtype=np.float64)
        if q.ndim == 2:
            return np.array([self.fkine(r) for r in q])
        T = np.eye(4)
        for i in range(self.n):
            T = T @ twist_exp(self.S[i], q[i])
        return T @ self.T0

    def _twist_jacobian(self, q):
        cols, T = [], np.eye(4)
        for i in range(self.n):
            cols.append(Ad(T) @ self.S[i])
            T = T @ twist_exp(self.S[i], q[i])
        return np.column_stack(cols) if cols else np.zeros((6, 0)), T @ self.T0

    def jacob0(self, q):
        """PoERobot.jacob0 (PoERobot.py:230-250): velocity-twist Jacobian converted to spatial velocity at the end-effector."""
        q = np.asarray(q, dtype=np.float64)
        if q.ndim == 2:
            return np.array([self.jacob0(r) for r in q])
        J, T = self._twist_jacobian(q)
        Jsv = np.eye(6)
        Jsv[:3, 3:] = -skew(T[:3, 3])
        return Jsv @ J

    def jacobe(self, q):
        """PoERobot.jacobe (PoERobot.py:252-270): Ad(T^-1) J."""
        q = np.asarray(q, dtype=np.float64)
        if q.ndim == 2:
            return np.array([self.jacobe(r) for r in q])
        J, T = self._twist_jacobian(q)
        return Ad(se3_inv(T)) @ J

    # -------------------------------------------------------------------------------------------- the reference's lowering
    def kinds(self):
        """'R' / 'P' per joint: PoERevolute twists have w != 0, PoEPrismatic w = 0."""
        return ["P" if np.linalg.norm(s[3:]) == 0.0 else "R" for s in self.S]

    @staticmethod
    def _elementary_list(T):
        """tx ty tz Rz(yaw) Ry(pitch) Rx(roll) of a transform with the near-zero ones removed (PoERobot.py:102-112, :304-314)."""
        rpy = tr2rpy_zyx(T[:3, :3])
        items = [("tx", T[0, 3]), ("ty", T[1, 3]), ("tz", T[2, 3]), ("Rz", rpy[2]), ("Ry", rpy[1]), ("Rx", rpy[0])]
        return [(a, float(e)) for a, e in items if not np.isclose(e, 0.0)]

    def _world_frame(self, S, kind):
        """PoELink._ets_world (PoERobot.py:42-121) followed by Link.Ts (Link.py:226-244): the frame the reference attaches to a
        twist -- z along the screw axis, origin at the axis point nearest the base origin, x towards that point -- AFTER its
        round trip through the elementary-transform list."""
        v, w = S[:3], S[3:]
        ex, ez = np.array([1.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.0])
        if kind == "P":
            a_vec, n_vec, t_vec = v, ex, np.zeros(3)
        elif kind == "R":
            pp = np.cross(w, v)
            n_vec = ex if np.isclose(np.linalg.norm(pp), 0.0) else pp / np.linalg.norm(pp)
            a_vec, t_vec = w, pp
        else:
            n_vec, a_vec, t_vec = ex, ez, v
        o_vec = np.cross(a_vec, n_vec)
        T = np.eye(4)
        T[:3, :3] = oa2r(o_vec, a_vec)
        T[:3, 3] = t_vec
        Ts = np.eye(4)
        for a, e in self._elementary_list(T):
            Ts = Ts @ chains.elementary(a, e)
        return Ts

    def lowered(self):
        """PoERobot._update_ets (PoERobot.py:272-318) -> the item list oracle.chains.Chain takes (the robot's ets())."""
        kinds = self.kinds()
        world = [np.eye(4)] + [self._world_frame(self.S[i], kinds[i]) for i in range(self.n)] + [self.T0]
        items = []
        for i in range(1, self.n + 2):
            items += self._elementary_list(se3_inv(world[i - 1]) @ world[i])
            if i <= self.n:
                items.append(("Rz",) if kinds[i - 1] == "R" else ("tz",))
        return items

    def chain(self):
        return chains.Chain(self.lowered(), name="PoE-lowered")


# ------------------------------------------------------------------------------------------------ the reference test's robots
def test_robot_2rpr():
    """tests/test_PoERobot.py:16-25."""
    S = [unit_revolute([0, 0, 1], [0, 0, 0]), unit_revolute([0, 1, 0], [0, 0, 0.2]), unit_prismatic([0, 1, 0]),
         unit_revolute([0, -1, 0], [0.2, 0, 0.5])]
    T0 = np.array([[1, 0, 0, 0.3], [0, 0, -1, 0], [0, 1, 0, 0.5], [0, 0, 0, 1.0]])
    return PoE(S, T0), np.array([-1.3, 0, 2.5, -1.7])


def trnorm(T):
    """spatialmath.base.trnorm (tests/test_PoERobot.py:65): o = R[:,1], a = R[:,2]; n = o x a; o = a x n; unit columns."""
    o, a = T[:3, 1], T[:3, 2]
    n = np.cross(o, a)
    o = np.cross(a, n)
    unit = lambda x: x / np.linalg.norm(x)
    X = np.eye(4)
    X[:3, :3] = np.stack((unit(n), unit(o), unit(a)), axis=1)
    X[:3, 3] = T[:3, 3]
    return X


def test_robot_3rp():
    """tests/test_PoERobot.py:41-68."""
    unit = lambda x: np.asarray(x, dtype=np.float64) / np.linalg.norm(x)
    S = [unit_revolute([0, 0, 1], [0, 0, 0]),
         unit_revolute(unit([-0.635, 0.495, 0.592]), [-0.152, -0.023, -0.144]),
         unit_revolute(unit([-0.280, 0.790, 0.544]), [-0.300, -0.003, -0.150]),
         unit_prismatic(unit([-0.280, 0.790, 0.544]))]
    T0 = trnorm(np.array([[0.2535, -0.5986, 0.7599, 0.2938], [-0.8063, 0.3032, 0.5078, -0.0005749],
                          [-0.5344, -0.7414, -0.4058, 0.08402], [0, 0, 0, 1.0]]))
    return PoE(S, T0), np.array([-1.3, -0.4, 2.5, -1.7])


test_robot_2rpr.__test__ = False
test_robot_3rp.__test__ = False

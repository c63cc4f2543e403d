"""RobotKinematics -- the robot-level kinematics surface, shared by every robot class of this package.

Mirrors reference robot/RobotKinematics.py (fkine :28-97, jacob0 :99-158, jacobe :160-219, hessian0 :222-333, hessiane
:335-445, partial_fkine0 :447-500, jacob0_analytical :502-570, ik_LM :576-746, ik_NR :748-879, ik_GN :881-1027, ikine_LM
:1029-1226, ikine_NR :1228-1368, ikine_GN :1370-1525, ikine_QP :1527-1735) and the three differential-kinematics methods that
live on Robot (manipulability robot/Robot.py:701-850, jacob0_dot :964-1098, jacobm :1101-1235): every one of them is
`self.ets(start, end).<method>(...)` in the reference, and is exactly that here -- the ETS object launches the kernel.  The
signatures are the reference's (same names, same order, same defaults), so an unknown keyword is a TypeError, never ignored.

A class using the mixin provides `ets(start=None, end=None) -> rtbhip.ETS` and `n`; optionally `base` (4x4, applied by fkine
only: Robot.jacob0 never sees the base, RobotKinematics.py:158) and `tool` (4x4 appended when the caller passes none).
"""
import numpy as np


class RobotKinematics:
    base = None
    tool = None

    # ------------------------------------------------------------ the path of a call
    @staticmethod
    def _link_key(x):
        return x if x is None or isinstance(x, (str, int)) else id(x)

    def _path(self, start, end):
        """self.ets(start, end), kept per (start, end) so that repeated calls reuse one device chain table, and widened to the
        ROBOT's q: a branch of a tree robot is evaluated on the robot-wide joint vector (robot/Robot.py:1974-1981)."""
        cache = self.__dict__.setdefault("_path_cache", {})
        key = (self._link_key(start), self._link_key(end))
        hit = cache.get(key)
        if hit is None:
            e = self.ets(start, end)
            n = int(getattr(self, "n", 0) or 0)
            if e.n and e.q_width < n <= 256:
                e.q_width = n
            hit = cache[key] = (e, start, end)          # the Link objects stay referenced: an id() is never reused under the key
        return hit[0]

    def _paths_changed(self):
        self.__dict__.pop("_path_cache", None)

    def _tool(self, tool):
        return self._kin_tool() if tool is None else tool

    def _kin_base(self):
        """The base transform fkine applies (None: identity).  DHRobot overrides both hooks: its ets() already holds base and tool."""
        return self.base

    def _kin_tool(self):
        """The tool a pass-through appends when the caller names none: NONE.  The reference's RobotKinematics hands `ets(start, end)` only
        the caller's `tool` (RobotKinematics.py:94, 158) and never reads `robot.tool` -- and the IK entry points solve the same bare chain,
        so that fkine(ik_LM(Tep).q) == Tep whatever `robot.tool` holds.  (A DHRobot's tool lives inside its ets(), a URDF robot's gripper
        tool inside ets(None): robot/BaseRobot.py:1610-1616.)"""
        return None

    # ------------------------------------------------------------ forward / differential kinematics
    def fkine(self, q, end=None, start=None, tool=None, include_base=True):
        """(4,4) or (N,4,4): RobotKinematics.py:28-97 (the reference wraps the same array in spatialmath.SE3)."""
        return self._path(start, end).fkine(q, base=self._kin_base(), tool=self._tool(tool), include_base=include_base)

    def jacob0(self, q, end=None, start=None, tool=None):
        return self._path(start, end).jacob0(q, tool=self._tool(tool))

    def jacobe(self, q, end=None, start=None, tool=None):
        return self._path(start, end).jacobe(q, tool=self._tool(tool))

    def hessian0(self, q=None, end=None, start=None, J0=None, tool=None):
        return self._path(start, end).hessian0(q, J0=J0, tool=self._tool(tool))

    def hessiane(self, q=None, end=None, start=None, Je=None, tool=None):
        return self._path(start, end).hessiane(q, Je=Je, tool=self._tool(tool))

    def partial_fkine0(self, q, n=3, end=None, start=None):
        return self._path(start, end).partial_fkine0(q, n=n, tool=self._kin_tool())

    def jacob0_analytical(self, q, representation="rpy/xyz", end=None, start=None, tool=None):
        return self._path(start, end).jacob0_analytical(q, representation=representation, tool=self._tool(tool))

    def manipulability(self, q=None, J=None, end=None, start=None, method="yoshikawa", axes="all"):
        """robot/Robot.py:701-905.  With J= (a finished Jacobian, (6,n) or a batch (N,6,n)) the measure is a pure function of it (:896) and
        runs on the device from the caller's array (rtbhip_manipulability_from_jacobian); with q the Jacobian never leaves the registers."""
        if J is not None:
            from .et import manipulability_from_jacobian
            return manipulability_from_jacobian(J, method=method, axes=axes)
        return self._path(start, end).manipulability(q, method=method, axes=axes, tool=self._kin_tool())

    def jacobm(self, q=None, J=None, H=None, end=None, start=None, axes="all"):
        """robot/Robot.py:1101-1235.  J= / H= (finished Jacobian, optionally the Hessian to go with it) are served from the caller's arrays
        (rtbhip_jacobm_from_jacobian; H alone needs q for the Jacobian, as in the reference :1194-1199)."""
        if q is None and J is None:
            q = np.copy(self.q)                       # the robot's stored configuration (robot/Robot.py:1195-1198)
        if J is not None or H is not None:
            from .et import jacobm_from_jacobian
            if J is None:
                J = self._path(start, end).jacob0(q, tool=self._kin_tool())
            return jacobm_from_jacobian(J, H=H, axes=axes)
        return self._path(start, end).jacobm(q, axes=axes, tool=self._kin_tool())

    def jacob0_dot(self, q, qd, J0=None, representation=None):
        """robot/Robot.py:964-1098 (no start / end there either)."""
        return self._path(None, None).jacob0_dot(q, qd, J0=J0, representation=representation, tool=self._kin_tool())

    # ------------------------------------------------------------ inverse kinematics
    def ik_LM(self, Tep, end=None, start=None, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, k=1.0,
              method="chan", **batch):
        return self._path(start, end).ik_LM(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits,
                                            k=k, method=method, **self._batch_only(batch))

    def ik_NR(self, Tep, end=None, start=None, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, pinv=True,
              pinv_damping=0.0, **batch):
        return self._path(start, end).ik_NR(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits,
                                            pinv=pinv, pinv_damping=pinv_damping, **self._batch_only(batch))

    def ik_GN(self, Tep, end=None, start=None, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, pinv=True,
              pinv_damping=0.0, **batch):
        return self._path(start, end).ik_GN(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits,
                                            pinv=pinv, pinv_damping=pinv_damping, **self._batch_only(batch))

    @staticmethod
    def _batch_only(kw):
        """The one keyword this backend adds to the C-solver calls: `seed`, the key of the counter-based restart generator (the
        reference draws from an unseeded std::rand, core/ik.cpp:293)."""
        extra = set(kw) - {"seed"}
        if extra:
            raise TypeError("unexpected keyword argument(s): %s" % ", ".join(sorted(extra)))
        return kw

    def ikine_LM(self, Tep, end=None, start=None, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 k=1.0, method="chan", kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        return self._path(start, end).ikine_LM(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits,
                                               seed=seed, k=k, method=method, kq=kq, km=km, ps=ps, pi=pi, **kwargs)

    def ikine_NR(self, Tep, end=None, start=None, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 pinv=False, kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        return self._path(start, end).ikine_NR(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits,
                                               seed=seed, pinv=pinv, kq=kq, km=km, ps=ps, pi=pi, **kwargs)

    def ikine_GN(self, Tep, end=None, start=None, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 pinv=False, kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        return self._path(start, end).ikine_GN(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits,
                                               seed=seed, pinv=pinv, kq=kq, km=km, ps=ps, pi=pi, **kwargs)

    def ikine_QP(self, Tep, end=None, start=None, q0=None, ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None,
                 kj=1.0, ks=1.0, kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        return self._path(start, end).ikine_QP(Tep, q0=q0, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits,
                                               seed=seed, kj=kj, ks=ks, kq=kq, km=km, ps=ps, pi=pi, **kwargs)


def as_se3(T, what="T"):
    """4x4 float64 from an ndarray or a spatialmath-like object with `.A`."""
    if T is None:
        return None
    if hasattr(T, "A") and not isinstance(T, np.ndarray):
        T = T.A
    T = np.asarray(T, dtype=np.float64)
    if T.shape != (4, 4):
        raise ValueError("%s must be a 4x4 SE(3) matrix" % what)
    return T

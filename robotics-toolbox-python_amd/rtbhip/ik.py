"""The solver classes of the reference's robot/IK.py as front ends of the device solvers: `IK_NR`, `IK_GN`, `IK_LM`, `IK_QP`
(robot/IK.py:579-763, 1020-1220, 766-1017, 1222-1520) with the constructor arguments of `IKSolver.__init__` (:149-172) and
`solve(ets, Tep, q0)` (:174-290).  In the reference `solve` runs the Python search loop around the class's `step`; here the whole
loop -- restarts, steps, limit checks, for every target of a batch -- is one kernel (csrc/ik_kernels.hip), reached through the
`ETS.ikine_*` method of the same flavour.  `error` (:369-401) is the batched angle-axis kernel.  `step` is not exposed: it only
exists fused into the device loop."""
import numpy as np

from ._lib import as_numeric, is_torch
from .et import IKSolution, angle_axis

__all__ = ["IKSolver", "IK_NR", "IK_GN", "IK_LM", "IK_QP", "IKSolution"]


class IKSolver:
    """Common part of the solver classes (reference IKSolver robot/IK.py:103-172): the search limits, the tolerance on
    E = 0.5 e^T We e, the Cartesian mask (We = diag(mask)), joint-limit rejection and the seed of the restart generator."""

    _flavour = None

    def __init__(self, name="IK Solver", ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None):
        self.name = name
        self.slimit = slimit
        self.ilimit = ilimit
        self.tol = tol
        self.seed = seed
        mask = np.ones(6) if mask is None else np.asarray(mask, dtype=np.float64).reshape(6)
        self.We = np.diag(mask)
        self.joint_limits = joint_limits

    def _extra(self):
        return {}

    def solve(self, ets, Tep, q0=None):
        """IKSolution for one pose, or for every pose of a (N,4,4) stack / SE3 sequence (robot/IK.py:174-290)."""
        if self._flavour is None:
            raise NotImplementedError("IKSolver is abstract: use IK_NR, IK_GN, IK_LM or IK_QP")
        fn = getattr(ets, self._flavour)
        return fn(Tep, q0=q0, ilimit=self.ilimit, slimit=self.slimit, tol=self.tol, mask=np.diag(self.We).copy(),
                  joint_limits=self.joint_limits, seed=self.seed, **self._extra())

    def error(self, Te, Tep):
        """(e, E): the angle-axis error of `Te` against `Tep` and E = 0.5 e^T We e (robot/IK.py:369-401)."""
        Te = Te.A if hasattr(Te, "A") and not isinstance(Te, np.ndarray) and not is_torch(Te) else Te
        Tep = Tep.A if hasattr(Tep, "A") and not isinstance(Tep, np.ndarray) and not is_torch(Tep) else Tep
        e = angle_axis(Te, Tep)
        ee = e.detach().cpu().numpy() if is_torch(e) else np.asarray(e)
        E = 0.5 * np.einsum("...i,ij,...j->...", ee, self.We, ee)
        return e, (float(E) if np.ndim(E) == 0 else E)

    def step(self, ets, Tep, q):
        raise NotImplementedError("the step of %s exists only fused into the device search loop (csrc/ik_device.h)" % type(self).__name__)


class _NullSpace(IKSolver):
    def __init__(self, name, ilimit, slimit, tol, mask, joint_limits, seed, kq, km, ps, pi, kwargs):
        if kwargs:
            raise TypeError("%s() got an unexpected keyword argument '%s'" % (type(self).__name__, sorted(kwargs)[0]))
        super().__init__(name=name, ilimit=ilimit, slimit=slimit, tol=tol, mask=mask, joint_limits=joint_limits, seed=seed)
        self.kq, self.km, self.ps, self.pi = kq, km, ps, pi

    def _null(self):
        return dict(kq=self.kq, km=self.km, ps=self.ps, pi=self.pi)


class IK_NR(_NullSpace):
    """Newton-Raphson: q += pinv(J) e  (reference IK_NR robot/IK.py:579-763)."""
    _flavour = "ikine_NR"

    def __init__(self, name="IK Solver", ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None, pinv=False,
                 kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        super().__init__(name, ilimit, slimit, tol, mask, joint_limits, seed, kq, km, ps, pi, kwargs)
        self.pinv = pinv

    def _extra(self):
        return dict(pinv=self.pinv, **self._null())


class IK_GN(IK_NR):
    """Gauss-Newton (reference IK_GN robot/IK.py:1020-1220; its step is the same pinv(J) e)."""
    _flavour = "ikine_GN"


class IK_LM(_NullSpace):
    """Levenberg-Marquardt, Chan / Wampler / Sugihara damping (reference IK_LM robot/IK.py:766-1017)."""
    _flavour = "ikine_LM"

    def __init__(self, name="IK Solver", ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None, k=1.0,
                 method="chan", kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        super().__init__(name, ilimit, slimit, tol, mask, joint_limits, seed, kq, km, ps, pi, kwargs)
        if method.lower().startswith("sugi"):
            method = "sugihara"
        elif method.lower().startswith("wamp"):
            method = "wampler"
        else:
            method = "chan"                                  # robot/IK.py:962-969: anything else is Chan
        self.k, self.method = k, method

    def _extra(self):
        return dict(k=self.k, method=self.method, **self._null())


class IK_QP(_NullSpace):
    """Quadratic-programming step (reference IK_QP robot/IK.py:1222-1520; note the class default kj = 0.01)."""
    _flavour = "ikine_QP"

    def __init__(self, name="IK Solver", ilimit=30, slimit=100, tol=1e-6, mask=None, joint_limits=True, seed=None, kj=0.01, ks=1.0,
                 kq=0.0, km=0.0, ps=0.0, pi=0.3, **kwargs):
        super().__init__(name, ilimit, slimit, tol, mask, joint_limits, seed, kq, km, ps, pi, kwargs)
        self.kj, self.ks = kj, ks

    def _extra(self):
        return dict(kj=self.kj, ks=self.ks, **self._null())

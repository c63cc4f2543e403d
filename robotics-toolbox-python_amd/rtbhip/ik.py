"""The solver classes of the reference's robot/IK.py as front ends of the device solvers: `IK_NR`, `IK_GN`, `IK_LM`, `IK_QP`
(robot/IK.py:579-763, 1020-1220, 766-1017, 1222-1520) with the constructor arguments of `IKSolver.__init__` (:149-172) and
`solve(ets, Tep, q0)` (:174-290).  In the reference `solve` runs the Python search loop around the class's `step`; here the whole
loop -- restarts, steps, limit checks, for every target of a batch -- is one kernel (csrc/ik_kernels.hip), reached through the
`ETS.ikine_*` method of the same flavour.  `error` (:369-401) is the batched angle-axis kernel.  `step(ets, Tep, q)` (:994-1017 and its
siblings) is ONE iteration of the same device code -- a call of the search kernel with ilimit = slimit = 1 and a tolerance nothing meets -- for
one configuration or a batch.  A subclass that OVERRIDES `step` gets what the reference gives it: `solve` then runs the reference's Python
loop (:297-367: searches, steps, wrap, joint-limit check) around the user's step on the host."""
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
        if type(self).step is not getattr(_builtin_step_owner(type(self)), "step", None):
            return self._solve_with_user_step(ets, Tep, q0)          # a user's own step: the reference's loop around it, on the host
        if self._flavour is None:
            raise NotImplementedError("IKSolver is abstract: use IK_NR, IK_GN, IK_LM or IK_QP, or subclass it with a step()")
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
        """One iteration of this solver from `q` towards `Tep`: (E, q_new) with E the error BEFORE the step, as the reference's `step`
        (robot/IK.py:994-1017 IK_LM, :732-763 IK_NR, :1190-1220 IK_GN, :1411-1520 IK_QP).  One configuration ((n,) q, (4,4) Tep: a float and an
        (n,) array; an ndarray q is also updated in place, as there) or a batch ((N,n), (N,4,4) or one pose for all).  Served by the device
        search kernel asked for exactly one step: ilimit = slimit = 1, tol = 0 (no E is below it, so nothing is wrapped or limit-checked)."""
        if self._flavour is None:
            raise NotImplementedError("IKSolver is abstract: its subclasses supply step()")
        Tep = Tep.A if hasattr(Tep, "A") and not isinstance(Tep, np.ndarray) and not is_torch(Tep) else Tep
        qa = q.detach().cpu().numpy() if is_torch(q) else as_numeric(q, "q")
        single = qa.ndim == 1
        q2 = np.ascontiguousarray(qa.reshape(1, -1) if single else qa, dtype=np.float64)
        T2 = np.asarray(Tep.detach().cpu().numpy() if is_torch(Tep) else Tep, dtype=np.float64)
        if T2.ndim == 2:
            T2 = np.broadcast_to(T2, (q2.shape[0], 4, 4))
        sol = getattr(ets, self._flavour)(np.ascontiguousarray(T2), q0=q2, ilimit=1, slimit=1, tol=0.0, mask=np.diag(self.We).copy(), joint_limits=False,
                                          seed=self.seed, **self._extra())
        qn = np.asarray(sol.q, dtype=np.float64).reshape(q2.shape)
        E = np.asarray(sol.each["residual"] if sol.each else [sol.residual], dtype=np.float64).reshape(-1)
        if single:
            if isinstance(q, np.ndarray) and q.dtype == np.float64:
                q[...] = qn[0]                                   # the reference's `q[ets.jindices] += ...`
            return float(E[0]), qn[0]
        return E, qn

    def _check_jl(self, ets, q):
        """robot/IK.py:403-430: every joint inside its limits"""
        ql = np.asarray(ets.qlim, dtype=np.float64)
        return bool(np.all(q >= ql[0]) and np.all(q <= ql[1]))

    def _random_q(self, ets, rng, k):
        ql = np.asarray(ets.qlim, dtype=np.float64)
        return rng.uniform(ql[0], ql[1], (k, ets.n))                 # robot/IK.py:432-470

    def _solve_with_user_step(self, ets, Tep, q0):
        """The reference's loop (robot/IK.py:174-367) around a subclass's own `step(ets, Tep, q) -> (E, q)`: start vectors (the caller's rows
        first, then uniform draws inside the limits, :222-240), up to slimit searches of up to ilimit steps, a success wrapped into [-pi, pi)
        and checked against the joint limits, numpy.linalg.LinAlgError abandoning a search.  Host-side: it is the user's Python that runs."""
        if is_torch(Tep):
            Tep = Tep.detach().cpu().numpy()
        Tep = Tep.A if hasattr(Tep, "A") and not isinstance(Tep, np.ndarray) else np.asarray(Tep, dtype=np.float64)
        if isinstance(Tep, (list, tuple)):                              # spatialmath: a multi-valued SE3's .A is a list of 4x4
            Tep = np.asarray(Tep, dtype=np.float64)
        rng = np.random.default_rng(self.seed)
        if q0 is None:
            starts = self._random_q(ets, rng, self.slimit)
        else:
            q0 = np.atleast_2d(np.asarray(q0, dtype=np.float64))
            starts = np.vstack([q0, self._random_q(ets, rng, max(self.slimit - q0.shape[0], 0))]) if q0.shape[0] < self.slimit else q0

        def one(T):
            total_i, found_with_limits, linalg_error, E, q = 0, False, 0, 0.0, starts[0]
            for search in range(self.slimit):
                q = starts[search].copy()
                i = 0
                while i < self.ilimit:
                    i += 1
                    try:
                        E, q = self.step(ets, T, q)
                        q = np.asarray(q, dtype=np.float64)
                    except np.linalg.LinAlgError:
                        linalg_error += 1
                        break
                    if E < self.tol:
                        q = (q + np.pi) % (2 * np.pi) - np.pi
                        if not self._check_jl(ets, q) and self.joint_limits:
                            found_with_limits = True
                            break
                        return IKSolution(q=q, success=True, iterations=total_i + i, searches=search + 1, residual=E, reason="Success")
                total_i += i
            reason = "iteration and search limit reached"
            if linalg_error:
                reason += ", %d numpy.LinAlgError encountered" % linalg_error
            if found_with_limits:
                reason += ", solution found but violates joint limits"
            return IKSolution(q=q, success=False, iterations=total_i, searches=self.slimit, residual=E, reason=reason)
        if Tep.ndim == 3:
            # the reference's `traj` branch (robot/IK.py:262-287): ONE solution object -- q (N, n), success = all, iterations / searches summed,
            # the smallest residual, the reason of the last failure -- with the per-target arrays beside it as the built-in path has them
            sols = [one(T) for T in Tep]
            failed = [x for x in sols if not x.success]
            return IKSolution(q=np.array([x.q for x in sols]).reshape(len(sols), ets.n), success=not failed, iterations=int(sum(x.iterations for x in sols)),
                              searches=int(sum(x.searches for x in sols)), residual=float(min([x.residual for x in sols], default=np.inf)),
                              reason=failed[-1].reason if failed else "",
                              each={"success": np.array([x.success for x in sols]), "iterations": np.array([x.iterations for x in sols]),
                                    "searches": np.array([x.searches for x in sols]), "residual": np.array([x.residual for x in sols], dtype=np.float64)})
        if Tep.shape != (4, 4):
            raise ValueError("Tep must be a 4x4 SE3 matrix")                # robot/IK.py:256-257
        return one(Tep)


def _builtin_step_owner(cls):
    """the nearest class of this module in cls's ancestry: its `step` is the built-in one (a subclass that defines its own differs from it)"""
    for c in cls.__mro__:
        if c.__module__ == __name__:
            return c
    return IKSolver


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

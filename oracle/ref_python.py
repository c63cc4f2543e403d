"""oracle/ref_python.py -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Runs the reference's OWN Python IK solvers -- /root/reference/src/roboticstoolbox/robot/IK.py, unmodified,
loaded from where it lies -- in a container where `import roboticstoolbox` is impossible (spatialmath-python,
a pyproject dependency, is absent and there is no network).  IK.py needs exactly four things from the outside
(IK.py:7-13, 218-238, 398, 568, 747-750, 994-1007):

  * `spatialmath.SE3`                     only for `isinstance(Tep, SE3)` / `.A` -> a 10-line stand-in below
  * `roboticstoolbox.tools.types`         loaded from the reference file itself
  * `rtb.angle_axis`                      the reference's tools/p_servo.py, loaded from the reference file; it calls
                                          `roboticstoolbox.fknm.Angle_Axis`, i.e. the reference's compiled extension
                                          (oracle/_ref, built unmodified by oracle/Makefile)
  * a duck-typed `ets`                    eval / jacob0 / jacobm / qlim / n / jindices / joints(): `DuckETS` below, every
                                          number coming out of the reference's compiled fknm; only `jacobm` (ETS.py:1669-1685,
                                          ten lines of NumPy over the reference's own J and H) is restated.
  * `qpsolvers.solve_qp` (IK_QP only, IK.py:16-21, 1497)   an absent optional third-party dependency (quadprog behind it).  The
                                          stand-in is oracle/qp.py: an exact solver that enumerates the active sets of the (few)
                                          inequality rows and solves each KKT system -- the minimiser of a strictly convex QP is
                                          unique, so it returns what quadprog returns to rounding.

Only tests/golden/make_golden.py (fixture generation, build container) and tests that are skipped where
/root/reference is absent import this module.  Nothing here can run on the GPU box.
"""
import importlib.util
import os
import sys
import types

import numpy as np

from . import ref_harness

REF_PKG = "/root/reference/src/roboticstoolbox"
_IK = None


def available():
    return os.path.isdir(REF_PKG) and ref_harness.available()


def _load_file(modname, path):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


class SE3:
    """Stand-in for spatialmath.SE3 as far as IK.py uses it (IK.py:246-254): isinstance, len, iteration, `.A`."""

    def __init__(self, A):
        A = np.asarray(A, dtype=float)
        self._data = [A] if A.ndim == 2 else list(A)

    def __len__(self):
        return len(self._data)

    def __iter__(self):
        return (SE3(a) for a in self._data)

    @property
    def A(self):
        return self._data[0] if len(self._data) == 1 else np.array(self._data)


def solve_qp(*args, **kw):
    """Stand-in for qpsolvers.solve_qp as IK_QP.step calls it (IK.py:1497): the exact enumerating solver of oracle/qp.py."""
    from . import qp
    return qp.solve_qp(*args, **kw)


def ik_module():
    """The reference's robot/IK.py as a module object (classes IK_LM, IK_NR, IK_GN, IK_QP, IKSolution, _calc_qnull ...)."""
    global _IK
    if _IK is not None:
        return _IK
    if not available():
        raise ImportError("needs /root/reference and oracle/_ref (build container only)")
    saved = {k: sys.modules.get(k) for k in ("spatialmath", "spatialmath.base", "qpsolvers", "roboticstoolbox", "roboticstoolbox.fknm",
                                             "roboticstoolbox.tools", "roboticstoolbox.tools.types",
                                             "roboticstoolbox.tools.p_servo", "roboticstoolbox.robot", "roboticstoolbox.robot.IK")}
    try:
        sm = types.ModuleType("spatialmath")
        sm.SE3 = SE3
        smb = types.ModuleType("spatialmath.base")
        # used only by p_servo's pure-Python fallback, which never runs here (Angle_Axis does not raise)
        smb.iszerovec = lambda v, tol=20: bool(np.linalg.norm(v) < tol * np.finfo(np.float64).eps)
        smb.norm = lambda v: float(np.linalg.norm(v))
        smb.isscalar = np.isscalar
        sm.base = smb
        sys.modules["spatialmath"], sys.modules["spatialmath.base"] = sm, smb
        qps = types.ModuleType("qpsolvers")
        qps.solve_qp = solve_qp
        sys.modules["qpsolvers"] = qps
        rtb = types.ModuleType("roboticstoolbox")
        rtb.__path__ = []
        sys.modules["roboticstoolbox"] = rtb
        fknm = ref_harness._load("fknm")
        sys.modules["roboticstoolbox.fknm"] = fknm
        rtb.fknm = fknm
        tools = types.ModuleType("roboticstoolbox.tools")
        tools.__path__ = []
        sys.modules["roboticstoolbox.tools"] = tools
        rtb.tools = tools
        tools.types = _load_file("roboticstoolbox.tools.types", os.path.join(REF_PKG, "tools", "types.py"))
        ps = _load_file("roboticstoolbox.tools.p_servo", os.path.join(REF_PKG, "tools", "p_servo.py"))
        rtb.angle_axis = ps.angle_axis
        rtb.angle_axis_python = ps.angle_axis_python
        robot = types.ModuleType("roboticstoolbox.robot")
        robot.__path__ = []
        sys.modules["roboticstoolbox.robot"] = robot
        _IK = _load_file("roboticstoolbox.robot.IK", os.path.join(REF_PKG, "robot", "IK.py"))
        _IK._p_servo = ps
    finally:
        # IK.py keeps its own references (`rtb`, `SE3`); the stand-ins must not leak into other importers
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return _IK


def angle_axis(Te, Tep):
    """`rtb.angle_axis` as IK.py sees it: p_servo.angle_axis -> the compiled Angle_Axis (fknm.cpp:112-162)."""
    return np.array(ik_module()._p_servo.angle_axis(np.asarray(Te, dtype=float), np.asarray(Tep, dtype=float)))


class _Joint:
    def __init__(self, jindex):
        self.jindex = jindex


class DuckETS:
    """What IK.py asks of an `rtb.ETS`, answered by the reference's compiled extension (an oracle.chains.Chain behind it)."""

    def __init__(self, ch):
        self.ref = ref_harness.RefETS(ch)
        self.ch = ch
        self.n = ch.n
        self.qlim = np.array(ch.qlim, dtype=float).reshape(2, ch.n)
        self.jindices = np.array([int(ch.jindex[i]) for i in range(ch.m) if int(ch.kind[i]) != 6])

    def joints(self):
        return [_Joint(int(j)) for j in self.jindices]

    def eval(self, q):                       # ETS.eval -> ETS_fkine(self._fknm, q, base, tool, include_base), ETS.py:1075-1077
        return np.array(self.ref.fknm.ETS_fkine(self.ref.cap, np.asarray(q, dtype=float), None, None, 1))

    def jacob0(self, q):                     # ETS.jacob0 -> ETS_jacob0(self._fknm, q, tool), ETS.py:1196-1198
        return np.array(self.ref.fknm.ETS_jacob0(self.ref.cap, np.asarray(q, dtype=float), None))

    def hessian0(self, q):                   # ETS.hessian0 -> ETS_hessian0(self._fknm, q, J0, tool), ETS.py:1385-1397
        q = np.asarray(q, dtype=float)
        return np.array(self.ref.fknm.ETS_hessian0(self.ref.cap, q, self.ref.fknm.ETS_jacob0(self.ref.cap, q, None), None))

    def manipulability(self, q):             # ETS.manipulability, method "yoshikawa", axes "all": ETS.py:1784-1791
        J = self.jacob0(q)
        if J.shape[0] == J.shape[1]:
            return abs(np.linalg.det(J))
        return np.sqrt(abs(np.linalg.det(J @ J.T)))

    def jacobm(self, q):                     # ETS.jacobm, ETS.py:1669-1685
        J = self.jacob0(q)
        H = self.hessian0(q)
        manipulability = self.manipulability(q)
        b = np.linalg.inv(J @ J.T)
        Jm = np.zeros((self.n, 1))
        for i in range(self.n):
            c = J @ H[i, :, :].T
            Jm[i, 0] = manipulability * (c.flatten("F")).T @ b.flatten("F")
        return Jm


def solve(solver_name, ch, Tep, q0, **kw):
    """IK_LM / IK_NR / IK_GN (`solver_name`) of the reference, `.solve(ets, Tep, q0)` -> (q, success, iterations, searches,
    residual).  q0: (n,) one start (the other slimit-1 are random) or (k, n) the first k starts (IK.py:230-238)."""
    ik = ik_module()
    solver = getattr(ik, solver_name)(**kw)
    sol = solver.solve(DuckETS(ch) if not isinstance(ch, DuckETS) else ch, np.asarray(Tep, dtype=float), q0)
    return np.array(sol.q, dtype=float), int(bool(sol.success)), int(sol.iterations), int(sol.searches), float(sol.residual)

"""oracle/qp.py -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

An exact solver for the small strictly convex quadratic programmes of the reference's IK_QP.step (robot/IK.py:1437-1500):

        min 1/2 x^T P x + q^T x     s.t.   A x = b,   G x <= h

The reference hands them to qpsolvers.solve_qp(..., solver="quadprog"), an optional third-party dependency that is absent here
(no network).  The minimiser of a strictly convex QP is unique, so any exact method returns what quadprog returns (to rounding);
this one is deliberately NOT the device's method (a primal-dual active set, csrc/ik_device.h): it ENUMERATES the candidate active
sets of the inequality rows (at most n rows are non-trivial -- one velocity damper per joint), solves the KKT system of each, and
keeps the one that is primal feasible with non-negative multipliers.  Exponential, which is fine for <= 16 rows.
Cross-checked against scipy.optimize (SLSQP) in tests/test_oracle_pins.py.
"""
import itertools

import numpy as np


def _kkt(P, q, A, b):
    n, m = P.shape[0], A.shape[0]
    K = np.block([[P, A.T], [A, np.zeros((m, m))]])
    sol = np.linalg.solve(K, np.concatenate((-q, b)))
    return sol[:n], sol[n:]


def solve_qp(P, q, G=None, h=None, A=None, b=None, lb=None, ub=None, solver=None, **kw):
    """Same call shape as qpsolvers.solve_qp as IK_QP.step uses it (IK.py:1497).  Returns x, or None when no KKT point is found
    (IK.py:1499-1500 turns None into LinAlgError("QP Unsolvable"))."""
    if lb is not None or ub is not None:
        raise NotImplementedError("oracle QP: bounds are passed as rows of G by the reference, never as lb / ub")
    P, q = np.asarray(P, dtype=float), np.asarray(q, dtype=float)
    n = P.shape[0]
    A = np.zeros((0, n)) if A is None else np.asarray(A, dtype=float).reshape(-1, n)
    b = np.zeros(0) if b is None else np.asarray(b, dtype=float).reshape(-1)
    if not (np.all(np.isfinite(P)) and np.all(np.isfinite(q)) and np.all(np.isfinite(A)) and np.all(np.isfinite(b))):
        return None
    rows = []
    if G is not None:
        G, h = np.asarray(G, dtype=float).reshape(-1, n), np.asarray(h, dtype=float).reshape(-1)
        if not (np.all(np.isfinite(G)) and np.all(np.isfinite(h))):
            return None
        for i in range(G.shape[0]):
            if np.any(G[i] != 0.0):
                rows.append(i)
            elif h[i] < 0.0:
                return None                    # 0 <= h violated: infeasible
    try:
        for k in range(len(rows) + 1):         # small active sets first: the usual answer has 0..2 active rows
            for act in itertools.combinations(rows, k):
                Aa = np.vstack([A] + [G[i:i + 1] for i in act]) if act else A
                ba = np.concatenate([b] + [h[i:i + 1] for i in act]) if act else b
                try:
                    x, lam = _kkt(P, q, Aa, ba)
                except np.linalg.LinAlgError:
                    continue
                if not np.all(np.isfinite(x)):
                    continue
                mu = lam[A.shape[0]:]          # multipliers of the active inequality rows: must be >= 0
                if np.any(mu < -1e-12 * (1.0 + np.abs(mu).max(initial=0.0))):
                    continue
                if rows and np.any(G[rows] @ x - h[rows] > 1e-12 * (1.0 + np.abs(h[rows]))):
                    continue
                return x
    except np.linalg.LinAlgError:
        return None
    return None

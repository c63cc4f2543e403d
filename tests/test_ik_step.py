"""`IK_LM.step` / `IK_NR.step` / `IK_GN.step` / `IK_QP.step` of rtbhip.ik -- ONE iteration of the device search kernel -- against the reference's own
`step` methods (robot/IK.py:994-1017, :732-763, :1190-1220, :1411-1520, executed unmodified: oracle/ref_classes.py on the reference's compiled
fknm): E before the step and the stepped q, for single configurations and batches, every damping rule; and a user subclass that overrides
`step`: `solve` then runs the reference's loop (:297-367) around it on the host, as the reference's `solve` would.  (Round 4's review: the
class was a shell whose step() raised.)"""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import ik as rik
from oracle import chains, ref_classes, ref_harness

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (ref_classes.available() and ref_harness.available()),
                                                  reason="needs oracle/_ref (the reference's compiled extension and byte-compiled classes)")]


def _mine():
    ets = rtbhip.models.Panda().ets()
    ets.qlim = chains.PANDA_QLIM
    return ets


@pytest.mark.parametrize("name,kw", [("IK_LM", dict(method="chan", k=1.0)), ("IK_LM", dict(method="wampler", k=0.01)), ("IK_LM", dict(method="sugihara", k=0.1)),
                                     ("IK_NR", dict(pinv=True)), ("IK_GN", dict(pinv=True)), ("IK_QP", dict(kj=0.01, ks=1.0))])
def test_step_equals_the_reference_s_step(name, kw):
    ns = ref_classes.load_reference()
    ref_ets = ref_classes.panda(ns)
    ets = _mine()
    rng = np.random.default_rng(7)
    qs = rng.uniform(chains.PANDA_QLIM[0], chains.PANDA_QLIM[1], (24, 7))
    q = np.clip(qs + 0.3 * rng.normal(size=qs.shape), chains.PANDA_QLIM[0], chains.PANDA_QLIM[1])
    Tep = np.asarray(ets.eval(qs))
    ref_solver = getattr(ns.IK, name)(**kw)
    solver = getattr(rik, name)(**kw)
    want = [ref_solver.step(ref_ets, Tep[i], q[i].copy()) for i in range(len(q))]
    # one configuration at a time: (float, (n,)); an ndarray q is stepped in place as in the reference
    for i in (0, 5):
        qi = q[i].copy()
        E, qn = solver.step(ets, Tep[i], qi)
        assert isinstance(E, float) and qn.shape == (7,)
        nt.assert_allclose(E, want[i][0], rtol=1e-12, atol=1e-15)
        nt.assert_allclose(qn, want[i][1], atol=1e-9)
        nt.assert_array_equal(qi, qn)
    # a batch
    E, qn = solver.step(ets, Tep, q.copy())
    assert E.shape == (24,) and qn.shape == (24, 7)
    nt.assert_allclose(E, [w[0] for w in want], rtol=1e-12, atol=1e-15)
    nt.assert_allclose(qn, np.array([w[1] for w in want]), atol=1e-9)


def test_a_user_step_is_looped_as_the_reference_loops_it():
    """a subclass with its own step (here: the built-in LM step with half the gain on the update) -- solve() runs the searches on the host"""
    calls = {"n": 0}

    class Half(rik.IK_LM):
        def step(self, ets, Tep, q):
            calls["n"] += 1
            E, qn = rik.IK_LM.step(self, ets, Tep, np.array(q, dtype=np.float64))
            return E, np.asarray(q) + 0.5 * (qn - np.asarray(q))

    ets = _mine()
    q_true = np.array([0.1, -0.4, 0.2, -2.0, 0.1, 1.8, 0.5])
    Tep = np.asarray(ets.eval(q_true))
    sol = Half(ilimit=60, slimit=3, seed=1).solve(ets, Tep, q0=q_true + 0.1)
    assert sol.success and sol.searches == 1 and 2 <= sol.iterations <= 60 and calls["n"] == sol.iterations
    assert np.abs(np.asarray(ets.eval(sol.q)) - Tep).max() < 2e-3 and sol.residual < 1e-6 and sol.reason == "Success"
    many = Half(ilimit=60, slimit=2, seed=1).solve(ets, np.stack([Tep, Tep]), q0=q_true + 0.1)
    # a stack of targets comes back as ONE solution, as the reference's `traj` branch returns it (robot/IK.py:262-287): q (N, n), success = all,
    # iterations / searches summed, the smallest residual
    assert many.q.shape == (2, ets.n) and many.success is True and many.searches == 2 and many.iterations == 2 * sol.iterations
    assert many.residual == sol.residual and np.array_equal(many.q[0], sol.q) and list(many.each["success"]) == [True, True]
    # the built-in classes keep the device loop: no host iteration
    calls["n"] = 0
    assert rik.IK_LM(seed=1).solve(ets, Tep, q0=q_true + 0.1).success and calls["n"] == 0
    with pytest.raises(NotImplementedError):
        rik.IKSolver().step(ets, Tep, q_true)

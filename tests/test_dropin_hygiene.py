"""Drop-in hygiene (round-2 review, items 7-9): a keyword the reference gives a meaning is honoured, one it ignores is refused --
never swallowed; the failure modes of the Python solvers are the reference's; idle staging memory can be handed back; tables can be
made resident up front so that hipGraph capture needs no warm-up call."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from oracle import chains, oracle


def test_dhrobot_refuses_what_the_reference_ignores():
    puma = rtbhip.models.DH.Puma560()
    q = puma.qn
    for call in (lambda: puma.fkine(q, end="link3"), lambda: puma.fkine(q, tool=np.eye(4)), lambda: puma.jacob0(q, end="x"),
                 lambda: puma.jacob0(q, start="x"), lambda: puma.jacobe(q, end="x"), lambda: puma.hessian0(q, end="x"),
                 lambda: puma.ets(end="link2"), lambda: puma.jacob0(q, bogus=1), lambda: puma.ik_LM(np.eye(4), end="x")):
        with pytest.raises(TypeError):
            call()
    with pytest.raises(ValueError):
        puma._half(np.zeros((6, 6)), "both")                 # robot/DHRobot.py:1139 "bad half specified"
    # None is what the reference's own pass-throughs hand over: accepted
    assert puma.ets(None, None) is puma.ets()


def test_solver_keywords_are_checked():
    ets = rtbhip.models.Panda().ets()
    for fn in (ets.ikine_LM, ets.ikine_NR, ets.ikine_GN, ets.ikine_QP):
        with pytest.raises(TypeError):
            fn(np.eye(4), tolerance=1e-3)                     # IKSolver.__init__ (robot/IK.py:149) has no such argument
    with pytest.raises(ValueError):
        ets.ikine_LM(np.eye(4), kq=0.1, pi=[0.3, 0.2])        # one influence distance per joint, or a scalar
    robot = rtbhip.models.Panda()
    with pytest.raises(TypeError):
        robot.fkine(np.zeros(7), bogus=1)
    with pytest.raises(TypeError):
        robot.ik_LM(np.eye(4), damping=3)


def test_ikine_without_pinv_on_a_redundant_arm_fails_like_the_reference():
    """numpy.linalg.inv of a 6x7 Jacobian raises LinAlgError in the first step of every search; the solver loop catches it
    (robot/IK.py:317-323) and, after slimit one-iteration searches, returns a failed IKSolution (:349-367)."""
    ets = rtbhip.models.Panda().ets()
    ets.qlim = chains.PANDA_QLIM
    Tep = oracle.fkine(chains.panda_ets(), np.array([[0, -0.3, 0, -2.2, 0, 2, 0.7]]))[0]
    for fn in (ets.ikine_NR, ets.ikine_GN):
        sol = fn(Tep, slimit=7, seed=11, pinv=False)
        assert sol.success is False and sol.iterations == 7 and sol.searches == 7 and sol.residual == 0.0
        assert "7 numpy.LinAlgError encountered" in sol.reason
        nt.assert_array_equal(sol.q, ets.ik_restart(11, 0, 6))          # the last search's (untouched) start vector
    sol = ets.ikine_NR(np.stack([Tep, Tep]), slimit=3, seed=2, pinv=False)
    assert sol.q.shape == (2, 7) and not sol.each["success"].any() and list(sol.each["iterations"]) == [3, 3]
    sol = ets.ikine_NR(Tep, q0=np.ones(7) * 0.1, slimit=1, pinv=False)
    nt.assert_array_equal(sol.q, np.ones(7) * 0.1)
    from oracle import ref_python
    if ref_python.available():                                            # the live reference, same start table
        starts = np.array([ets.ik_restart(11, 0, d) for d in range(7)])
        q, ok, it, se, E = ref_python.solve("IK_NR", chains.panda_ets(with_limits=True), Tep, starts, slimit=7, pinv=False)
        ours = ets.ikine_NR(Tep, slimit=7, seed=11, pinv=False)
        assert (ok, it, se, E) == (0, ours.iterations, ours.searches, ours.residual)
        nt.assert_array_equal(q, ours.q)


def test_trim_and_upload_entry_points_exist_and_fail_loudly_without_a_gpu():
    rtbhip.trim()                                                         # nothing cached: a no-op
    if rtbhip.device_count() == 0:
        with pytest.raises(rtbhip.RtbHipError):
            rtbhip.models.Panda().ets().upload()


# ------------------------------------------------------------------------------------------------ on the device
@pytest.mark.gpu
def test_gpu_dhrobot_half_and_supplied_pose():
    puma = rtbhip.models.DH.Puma560()
    rng = np.random.default_rng(0)
    for q in (puma.qn, rng.uniform(-2, 2, (33, 6))):
        J0, Je = puma.jacob0(q), puma.jacobe(q)
        nt.assert_array_equal(puma.jacob0(q, half="trans"), J0[..., :3, :])
        nt.assert_array_equal(puma.jacob0(q, half="rot"), J0[..., 3:, :])
        nt.assert_array_equal(puma.jacobe(q, half="trans"), Je[..., :3, :])
        assert puma.jacob0(q, half="trans").shape[-2:] == (3, 6)          # the reference returns (3,n) (robot/DHRobot.py:1188-1196)
        T = puma.fkine(q)
        nt.assert_allclose(puma.jacob0(q, T=T), J0, atol=1e-12)           # tr2jac(fkine(q)) @ jacobe(q) IS jacob0(q)
        nt.assert_allclose(puma.jacob0(q, T=T, half="rot"), J0[..., 3:, :], atol=1e-12)
        nt.assert_array_equal(puma.jacob0_analytical(q), J0)              # representation=None (robot/DHRobot.py:1257-1258)
    other = chains.elementary("Rz", 0.7) @ chains.elementary("Rx", -0.4)
    R6 = np.zeros((6, 6)); R6[:3, :3] = other[:3, :3]; R6[3:, 3:] = other[:3, :3]
    nt.assert_allclose(puma.jacob0(puma.qn, T=other), R6 @ puma.jacobe(puma.qn), atol=1e-13)     # a SUPPLIED pose is used
    import torch
    qd = torch.from_numpy(rng.uniform(-2, 2, (9, 6))).cuda()
    got = puma.jacob0(qd, T=puma.fkine(qd), half="trans")
    assert got.is_cuda
    nt.assert_allclose(got.cpu().numpy(), puma.jacob0(qd.cpu().numpy())[:, :3, :], atol=1e-12)


@pytest.mark.gpu
def test_gpu_capture_without_a_warm_up_call_after_upload():
    """rtbhip_chain_upload / rtbhip_dyn_upload make the tables resident first: the very first launches of fresh handles are then
    captured into a hipGraph (no eager call before the capture) and replay bit-equal to eager results."""
    import torch
    panda = rtbhip.ETS(list(rtbhip.models.Panda().ets()))                 # fresh handles: nothing has run on them
    panda.qlim = rtbhip.models.PANDA_QLIM
    arm = rtbhip.models.DH.Panda()
    rng = np.random.default_rng(5)
    N = 2048
    q = torch.from_numpy(rng.uniform(panda.qlim[0], panda.qlim[1], (N, 7))).cuda()
    qd = torch.from_numpy(rng.normal(size=(N, 7))).cuda()
    panda.upload()
    arm.upload()
    torch.cuda.synchronize()

    def step():
        T, J = panda.fkine_jacob0(q)
        tau = arm.rne(q, qd, qd)
        sol = panda.ik_LM(T, q0=q, seed=1)
        return T, J, tau, sol[0], sol[1]

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()
    g.replay()
    torch.cuda.synchronize()
    got = [x.clone() for x in outs]
    eager = step()
    torch.cuda.synchronize()
    for a, b in zip(got, eager):
        nt.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    ch = chains.panda_ets()
    nt.assert_allclose(got[0].cpu().numpy(), oracle.fkine(ch, q.cpu().numpy()), atol=1e-10)


@pytest.mark.gpu
def test_gpu_trim_returns_idle_staging_memory():
    import torch
    ets = rtbhip.models.Panda().ets()
    N = 400000
    J = np.random.default_rng(1).normal(size=(N, 6, 7))
    rtbhip.trim()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    H = rtbhip.hessian_from_jacobian(J)                                    # host path: stages 134 MB in, 941 MB out
    assert H.shape == (N, 7, 6, 7)
    free1, _ = torch.cuda.mem_get_info()
    # the cache is bounded (RTBHIP_DEVICE_CACHE_MB, default 512): the 941 MB output block went back when it was released
    assert free0 - free1 <= (512 + 64) << 20
    rtbhip.trim()
    free2, _ = torch.cuda.mem_get_info()
    assert free0 - free2 <= 64 << 20
    nt.assert_allclose(rtbhip.hessian_from_jacobian(J[:3]), H[:3], atol=0)

"""hipGraph capture of the device-pointer entry points: every call only enqueues kernels on the caller's stream
(tables are uploaded on first use, so one eager warm-up precedes the capture -- DESIGN.md 7), so a control loop's
fkine + Jacobian + inverse dynamics + IK step can be captured once and replayed with new inputs."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip


@pytest.mark.gpu
def test_capture_and_replay_kinematics_dynamics_ik():
    import torch
    panda = rtbhip.models.Panda().ets()
    panda.qlim = rtbhip.models.PANDA_QLIM
    arm = rtbhip.models.DH.Panda()
    rng = np.random.default_rng(3)
    N = 4096
    q = torch.from_numpy(rng.uniform(panda.qlim[0], panda.qlim[1], (N, 7))).cuda()
    qd = torch.from_numpy(rng.normal(size=(N, 7))).cuda()
    qdd = torch.from_numpy(rng.normal(size=(N, 7))).cuda()

    def step():
        T, J = panda.fkine_jacob0(q)
        H = panda.hessian0(q)
        tau = arm.rne(q, qd, qdd)
        M = arm.inertia(q)
        sol = panda.ik_LM(T, q0=q, seed=1)            # starts at the solution: first search, no restarts drawn
        return T, J, H, tau, M, sol[0], sol[1]

    eager0 = [x.clone() for x in step()]              # warm-up: uploads the tables, sizes the allocator pools
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(outs, eager0):
        nt.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    # new inputs in the captured buffers, replay, compare with an eager run on the same data
    q.copy_(torch.from_numpy(rng.uniform(panda.qlim[0], panda.qlim[1], (N, 7))))
    qd.copy_(torch.from_numpy(rng.normal(size=(N, 7))))
    g.replay()
    torch.cuda.synchronize()
    replayed = [x.clone() for x in outs]
    eager1 = step()
    torch.cuda.synchronize()
    for a, b in zip(replayed, eager1):
        nt.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    assert float((replayed[6] == 1).double().mean()) > 0.99

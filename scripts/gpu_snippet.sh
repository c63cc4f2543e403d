#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import numpy as np, torch, rtbhip
panda = rtbhip.models.Panda()
q = np.random.uniform(-np.pi, np.pi, (1_000_000, 7))
T = panda.fkine(q)
qd = torch.from_numpy(q).cuda()
T, J = panda.ets().fkine_jacob0(qd)
q_sol, ok, its, searches, E = panda.ets().ik_LM(T[:100_000])
arm = rtbhip.models.DH.Panda()
tau = arm.rne(qd, torch.zeros_like(qd), torch.zeros_like(qd))
M, C = arm.inertia(qd), arm.coriolis(qd, torch.randn_like(qd))
ur5 = rtbhip.urdf.load("UR5")
Ts, Js = rtbhip.fleet_fkine_jacob([ur5.ets(), panda.ets()], [q[:, :6].copy(), q])
torch.cuda.synchronize()
print("snippet ok", T.shape, J.shape, float(ok.float().mean()), tau.shape, M.shape, C.shape, Ts[0].shape, Js[1].shape)
PY

import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/robotics-toolbox-python_amd"]
import numpy as np, torch, rtbhip
ets = rtbhip.models.Panda().ets(); ets.qlim = rtbhip.models.PANDA_QLIM
T = ets.eval(torch.from_numpy(np.random.default_rng(1).uniform(ets.qlim[0], ets.qlim[1], (100000, 7))).cuda())
out = {}
for u in (0, 1):
    rtbhip.tune("ik_unit_we", u)
    out[u] = [x.cpu().numpy() for x in ets.ik_LM(T, seed=2)]
a, b = out[0], out[1]
print("rows with different ok/it/se:", int(((a[1] != b[1]) | (a[2] != b[2]) | (a[3] != b[3])).sum()))
same = (a[1] == b[1]) & (a[2] == b[2]) & (a[3] == b[3]) & (a[1] == 1)
print("max |dq| on rows with equal counts:", float(np.abs(a[0][same] - b[0][same]).max()), " max |dE|:", float(np.abs(a[4][same] - b[4][same]).max()))
print("rows with any q bit difference:", int((a[0] != b[0]).any(axis=1).sum()))

#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "" $R/robotics-toolbox-python_amd/lib/variants/hremap.so; do
for m in 0 2; do
RTBHIP_LIB=$v python - "$m" "${v:-identity}" <<'PY'
import sys; sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import numpy as np, torch, rtbhip
mode = int(sys.argv[1]); rtbhip.tune("hess_mode", mode)
e = rtbhip.models.Panda().ets()
q = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (1000000, 7))).cuda()
e.hessian0(q); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); e.hessian0(q); b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)
print("%-10s R=%d  avg %.4f ms  min %.4f" % (sys.argv[2][-10:], 4 if mode == 0 else 8, sum(ms) / len(ms), ms[0]))
PY
done; done; done

#!/bin/bash
export TMPDIR=/tmp
python - <<'PY'
import sys, time; sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import numpy as np, torch, rtbhip
ets = rtbhip.models.Panda().ets()
rng = np.random.default_rng(0)
for order, N in ((3, 100000), (4, 10000), (5, 1000)):
    q = torch.from_numpy(rng.uniform(-2, 2, (N, 7))).cuda()
    ets.partial_fkine0(q, order); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = ets.partial_fkine0(q, order); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    gb = out.numel() * 8 / 1e9
    print("order %d N=%d: %.3f ms (min), %.2f GB out -> %.2f TB/s" % (order, N, min(ts) * 1e3, gb, gb / min(ts) / 1e3))
PY

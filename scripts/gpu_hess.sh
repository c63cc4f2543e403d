#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import numpy as np, torch, rtbhip
ets = rtbhip.models.Panda().ets()
puma = rtbhip.models.DH.Puma560().ets()
rng = np.random.default_rng(0)
for e in (ets, puma):
    q = torch.from_numpy(rng.uniform(-3, 3, (1000003, e.n))).cuda()
    out = {}
    for mode in (0, 1, 2, 5):
        rtbhip.tune("hess_mode", mode)
        H = e.hessian0(q); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); e.hessian0(q); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        out[mode] = H
        print("n=%d hess_mode=%d avg %.4f ms min %.4f ms  %.0f GB/s" % (e.n, mode, sum(ms) / len(ms), ms[0], (8 * e.n + 48 * e.n * e.n) * q.shape[0] / ms[0] / 1e6))
    print("  modes agree:", bool(all((out[0] == out[m]).all() for m in out)))
PY

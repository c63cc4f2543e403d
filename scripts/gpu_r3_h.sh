#!/bin/bash
# Round 3, visit h: why does the fleet line's host loop take 2.6 ms per step against 1.45 ms of kernels?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
RTBHIP_BENCH_TRACE=1 timeout 300 python bench_extra.py --what fleet --no-cpu --steps 20 2>&1 | cut -c1-400 | head -6
python - <<'PY'
import sys, time, numpy as np, torch
sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import rtbhip
from rtbhip import urdf
N = 1000000
robots = [urdf.load(nm) for nm in urdf.FLEET16]
chs = [r.ets() for r in robots]
qs = [torch.from_numpy(np.random.default_rng(4 + i).uniform(-1, 1, (N, c.n))).cuda() for i, c in enumerate(chs)]
import cProfile, pstats
for _ in range(3): out = rtbhip.fleet_fkine_jacob(chs, qs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): out = rtbhip.fleet_fkine_jacob(chs, qs)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host per call %.3f ms, incl. drain %.3f ms" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): out = rtbhip.fleet_fkine_jacob(chs, qs)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
PY

#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time
sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import numpy as np, rtbhip
ets = rtbhip.models.Panda().ets()
q = np.random.default_rng(0).uniform(-np.pi, np.pi, (1000000, 7))
ets.fkine_jacob0(q[:1000])
for _ in range(3):
    t0 = time.perf_counter(); T, J = ets.fkine_jacob0(q); dt = time.perf_counter() - t0
    print("host-pointer path (pageable NumPy in/out, staged through the device): %.1f ms per 1e6 -> %.3g configs/s, %.2f GB/s over the 520 MB" % (dt * 1e3, 1e6 / dt, 0.52 / dt))
PY

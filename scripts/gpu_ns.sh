#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_00_gpu_parity.py -m gpu -x -q -k "ik" > gpurun_out/pytest_ns.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ns.log
tail -25 gpurun_out/pytest_ns.log
python - <<'PY'
import sys; sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import numpy as np, torch, rtbhip, time
ets = rtbhip.models.Panda().ets(); ets.qlim = rtbhip.models.PANDA_QLIM
rng = np.random.default_rng(1)
N = 100000
qs = torch.from_numpy(rng.uniform(ets.qlim[0], ets.qlim[1], (N, 7))).cuda()
Tep = ets.eval(qs)
for name, kw in (("plain", {}), ("kq=.1", dict(kq=0.1)), ("kq=km=.1", dict(kq=0.1, km=0.1))):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sol = ets.ikine_LM(Tep, seed=2, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("ikine_LM %-9s 1e5 targets: %.2f ms  success %.4f  iterations %d" % (name, dt * 1e3, sol.each["success"].mean(), sol.iterations))
PY

#!/bin/bash
# Round 3, visit n: coriolis with the per-row choice of scheme -- dynamics tests, then the dyn bench lines.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_dynamics_terms.py tests/test_00_gpu_parity.py -m gpu -q -x -k "dyn or coriolis or inertia or accel or Dynamics or terms" --timeout 600 2>&1 | tail -3
for rep in 1 2; do
timeout 300 python bench_extra.py --what dyn --no-cpu --steps 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['metric'][:50], 'avg %.4f min %.4f' % (d['kernel_avg_ms'], d['kernel_min_ms']))"
done

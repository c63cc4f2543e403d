#!/bin/bash
# Round 3, visit i: IK flat schedule with evidence-gated draws of later chunks: defaults, notebook setting, 1e6; bit-equality test; fleet line inside the full run.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_00_gpu_parity.py -m gpu -q -rf -k "ik" --timeout 600 2>&1 | tail -2
for rep in 1 2; do for f in 1 0; do
  timeout 300 python bench_extra.py --what ik --no-cpu --steps 12 --tune ik_flat=$f 2>/dev/null | python -c "
import sys,json
o=[]
for l in sys.stdin:
    d=json.loads(l); o.append('%s: %.4f/%.4f ms' % (d['metric'][18:62], d['kernel_avg_ms'], d['kernel_min_ms']))
print('ik_flat=$f', ' | '.join(o))"
done; done
timeout 900 python bench_extra.py --no-cpu 2>/dev/null | grep "mixed fleet: 16" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fleet inside the full run: ms_per_step %.3f kernel_avg %.3f min %.3f' % (d['ms_per_step'], d['kernel_avg_ms'], d['kernel_min_ms']))"

#!/bin/bash
# Round 3, visit d: whole -m gpu suite; k_rne with / without the p* = 0 shortcut (interleaved); IK pass-mask at the config-3 size; bench lines.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3d
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rf --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for rep in 1 2 3; do
for t in 1 0; do
  timeout 300 python bench_extra.py --what rne,dyn --no-cpu --steps 40 --tune rne_pszero=$t 2>/dev/null | python -c "
import sys,json
out=[]
for l in sys.stdin:
    d=json.loads(l); out.append('%s %.4f/%.4f' % (d['metric'].split('(')[1][:18], d['kernel_avg_ms'], d['kernel_min_ms']))
print('rne_pszero=$t', ' | '.join(out))"
done
done
for m in 3 1 7; do
  timeout 300 python bench_extra.py --what ik --no-cpu --steps 12 --tune ik_pass_mask=$m 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ik_pass_mask=$m', 'avg %.4f min %.4f ms' % (d['kernel_avg_ms'], d['kernel_min_ms']), '%.3g' % d['lm_iterations_per_s'])"
done
timeout 300 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-300 $O/bench_n1.json

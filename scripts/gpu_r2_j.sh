#!/bin/bash
# Round 2, visit j: acceleration-only passes for the columns of M(q) (inertia, accel): parity tests and timing
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_dynamics_terms.py tests/test_00_gpu_parity.py -m gpu -q --timeout 200 --tb=short -k "dyn or inertia or accel or coriolis or rne" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -v "Warning\|^  \|^$" $O/pytest_gpu.log | tail -12 | cut -c1-250
timeout 300 python bench_extra.py --what dyn,rne --no-cpu --steps 12 > $O/bench_dyn.jsonl 2>$O/err.txt
python - $O/bench_dyn.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); print("%-50s avg %.4f ms  min %.4f ms  frac %.3f" % (d["metric"][:50], d["kernel_avg_ms"], d["kernel_min_ms"], d["roofline"]["frac"]))
PY

#!/bin/bash
# Round 2, visit e: cross-wave sharing of IK search ranges on the GPU (equivalence tests, then timing at the config-3 size and
# around it), the rne loop trace with the collector out of the timed region
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2e
mkdir -p $O
timeout 900 python -m pytest tests/test_00_gpu_parity.py tests/test_03_python_ik_pins.py tests/test_hip_graph.py -m gpu -q --timeout 300 -k "ik or IK or graph" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|Timeout" $O/pytest_gpu.log | tail -8
echo "== ik at 1e5 targets"
for t in "--tune ik_share=0" "--tune ik_share=1" "--tune ik_share=1 --tune ik_fresh_pct=100" "--tune ik_share=1 --tune ik_fresh_pct=25" "--tune ik_share=0 --tune ik_fresh_pct=100"; do
  echo "  [$t]"; timeout 120 python bench_extra.py --what ik --no-cpu --steps 12 $t 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print("     n=%-8d avg %.3f ms min %.3f ms  %.3g it/s  ok %.4f mean it %.2f  frac %.3f" % (d["n"], d["kernel_avg_ms"], d["kernel_min_ms"], d["lm_iterations_per_s"], d["success_rate"], d["mean_iterations"], d["roofline"]["frac"]))'
done 2>&1 | tee $O/ik_share.txt
for n in 20000 50000 200000 400000; do for sh in 0 1; do echo "  n-ik $n share $sh"; timeout 120 python bench_extra.py --what ik --no-cpu --steps 8 --n-ik $n --tune ik_share=$sh 2>/dev/null | head -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); print("     avg %.3f ms min %.3f ms  %.3g it/s" % (d["kernel_avg_ms"], d["kernel_min_ms"], d["lm_iterations_per_s"]))'; done; done 2>&1 | tee -a $O/ik_share.txt
RTBHIP_BENCH_TRACE=1 python bench_extra.py --what rne --no-cpu --steps 30 2>&1 | cut -c1-420

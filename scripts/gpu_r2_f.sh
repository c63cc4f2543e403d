#!/bin/bash
# Round 2, visit f: ticket hand-over protocol for cross-wave sharing of IK search ranges (every step under its own short timeout)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2f
mkdir -p $O
pyline='
import json,sys
for l in sys.stdin:
    if not l.startswith("{"): continue
    d=json.loads(l); print("     n=%-8d avg %.3f ms min %.3f ms  %.3g it/s  ok %.4f mean it %.2f  frac %.3f" % (d["n"], d["kernel_avg_ms"], d["kernel_min_ms"], d["lm_iterations_per_s"], d["success_rate"], d["mean_iterations"], d["roofline"]["frac"]))'
echo "== ik at 1e5 targets" | tee $O/ik_share.txt
for t in "--tune ik_share=0" "--tune ik_share=1"; do
  echo "  [$t]"; timeout 100 python bench_extra.py --what ik --no-cpu --steps 12 $t 2>$O/err.txt | python -c "$pyline"; echo "  rc=${PIPESTATUS[0]}"
done 2>&1 | tee -a $O/ik_share.txt
for n in 20000 400000; do echo "  n-ik $n share 1"; timeout 60 python bench_extra.py --what ik --no-cpu --steps 8 --n-ik $n --tune ik_share=1 2>/dev/null | head -1 | python -c "$pyline"; done 2>&1 | tee -a $O/ik_share.txt
timeout 420 python -m pytest tests/test_00_gpu_parity.py -m gpu -q --timeout 150 --tb=short -k "cross_wave or fourteen" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -60 $O/pytest_gpu.log | cut -c1-300

#!/bin/bash
# Round 2, visit c: RNE variants A/B (old recursion / fused multiply-add chains / + operand prefetch at 2 and 3 waves per SIMD),
# IK scheduler knobs at the BASELINE config-3 size, phased schedule check, the rne loop-time question of visit a
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "rne or dyn or ik or IK or inertia or coriolis or accel" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
V=$R/robotics-toolbox-python_amd/lib/variants
for rep in 1 2; do
for v in "" $V/rne_base.so $V/rne_pf.so $V/rne_pf3.so; do
  echo "== rne $(basename ${v:-main_fma})"
  RTBHIP_LIB=$v python bench_extra.py --what rne --no-cpu --steps 30 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("  1.25e6: loop avg %.4f ms  min %.4f ms  %.4g triples/s" % (d["kernel_avg_ms"], d["kernel_min_ms"], d["value"]))'
  RTBHIP_LIB=$v python bench_extra.py --what rne --no-cpu --n-rne 10000000 --steps 10 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("  1e7   : loop avg %.4f ms  min %.4f ms  %.4g triples/s" % (d["kernel_avg_ms"], d["kernel_min_ms"], d["value"]))'
done; done 2>&1 | tee $O/rne_ab.txt
echo "== dyn (main_fma vs base)"
for v in "" $V/rne_base.so; do RTBHIP_LIB=$v python bench_extra.py --what dyn,tree --no-cpu --steps 10 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print("  %-55s avg %.4f min %.4f ms" % (d["metric"][:55], d["kernel_avg_ms"], d["kernel_min_ms"]))'; done 2>&1 | tee $O/dyn_ab.txt
echo "== ik knobs at 1e5 targets"
for t in "" "--tune ik_fresh_pct=80" "--tune ik_fresh_pct=60" "--tune ik_fresh_pct=35" "--tune ik_waves_per_cu=6" "--tune ik_waves_per_cu=4" "--tune ik_spec_policy=1" "--tune ik_phased=2" "--tune ik_fresh_pct=80 --tune ik_spec_policy=1"; do
  echo "  [$t]"; python bench_extra.py --what ik --no-cpu --steps 12 $t 2>/dev/null | head -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("     1e5: avg %.3f ms min %.3f ms  %.3g it/s  ok %.4f mean it %.2f" % (d["kernel_avg_ms"], d["kernel_min_ms"], d["lm_iterations_per_s"], d["success_rate"], d["mean_iterations"]))'
done 2>&1 | tee $O/ik_knobs.txt

#!/bin/bash
# Round 3, visit c: IK at config 3 -- plain schedule vs the flat schedule (one launch, chunked search ranges), per-wave occupancy counters.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_00_gpu_parity.py -m gpu -q -rf -k "ik" --timeout 600 > $O/pytest_ik.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_ik.log | tail -5
for rep in 1 2; do
for t in 0 1; do
  timeout 300 python bench_extra.py --what ik --no-cpu --steps 12 --tune ik_flat=$t 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ik_flat=$t', 'avg %.4f min %.4f ms' % (d['kernel_avg_ms'], d['kernel_min_ms']), d['success_rate'], d['mean_iterations'], '%.3g' % d['lm_iterations_per_s'])"
done
done
for cfg in "8 16" "4 8" "12 24"; do set -- $cfg
  timeout 300 python bench_extra.py --what ik --no-cpu --steps 12 --tune ik_flat=1 --tune ik_flat_l0=$1 --tune ik_flat_len=$2 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat l0=$1 len=$2', 'avg %.4f min %.4f ms' % (d['kernel_avg_ms'], d['kernel_min_ms']))"
done
for t in 0 1; do
  rm -f /tmp/ikstats.jsonl
  RTBHIP_IK_STATS=/tmp/ikstats.jsonl timeout 300 python bench_extra.py --what ik --no-cpu --steps 4 --tune ik_flat=$t > /dev/null 2>&1
  head -3 /tmp/ikstats.jsonl | tail -1 > $O/ikstats_flat$t.json
  python - $O/ikstats_flat$t.json <<'PY'
import json, sys, numpy as np
d = json.load(open(sys.argv[1]))
a = np.array(d["per_wave"], dtype=np.int64)
it, ps, ln, items = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
print("grid", d["grid"], "items", d["items"], "flat_chunks", d["flat_chunks"], "| wave iterations: mean %.1f max %d | passes mean %.1f | running lane-iterations %d (%.2f of the lane slots) | items started %d" % (it.mean(), it.max(), ps.mean(), ln.sum(), ln.sum() / (64.0 * it.sum()), items.sum()))
print("  histogram of wave iterations:", np.histogram(it, bins=[0, 40, 60, 80, 100, 120, 140, 160, 180, 200, 250, 400])[0].tolist())
PY
done

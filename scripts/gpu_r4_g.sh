#!/bin/bash
# Round 4, visit g: the new IK defaults (fresh share 100 %, first chunk automatic) -- the IK tests, the sustained A/B against the round-3 knobs,
# the per-wave occupancy counters of both, the bench line.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4g}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_00_gpu_parity.py tests/test_03_python_ik_pins.py tests/test_02_compat_shim.py -m gpu -q -rf --timeout 600 > $O/pytest_ik.log 2>&1; grep -E "passed|failed|FAILED" $O/pytest_ik.log | tail -5
IK_AB_SET=6 timeout 600 python scripts/ik_ab.py 100000 3 > $O/ik_ab6.jsonl 2> $O/ik_ab6.err; cut -c1-200 $O/ik_ab6.jsonl; tail -2 $O/ik_ab6.err
for v in "r3 --tune ik_flat_l0=4 --tune ik_fresh_pct=50" "r4"; do set -- $v; name=$1; shift
  rm -f /tmp/ikstats.jsonl
  RTBHIP_IK_STATS=/tmp/ikstats.jsonl timeout 300 python bench_extra.py --what ik --no-cpu --steps 4 --n-ik 100000 "$@" > /dev/null 2>&1
  head -3 /tmp/ikstats.jsonl | tail -1 > $O/ikstats_$name.json
  python - $O/ikstats_$name.json $name <<'PY'
import json, sys, numpy as np
d = json.load(open(sys.argv[1]))
a = np.array(d["per_wave"], dtype=np.int64)
it, ps, ln, items = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
useful = 7.504e6
slots = 64.0 * it.sum()
print(sys.argv[2], "grid", d["grid"], "items", d["items"], "flat_chunks", d["flat_chunks"], "| wave iterations: mean %.1f median %d p99 %d max %d | passes mean %.1f | items started %d" % (it.mean(), np.median(it), np.percentile(it, 99), it.max(), ps.mean(), items.sum()))
print("   lane slots %.3e = useful %.1f %% + discarded %.1f %% + idle %.1f %%" % (slots, 100 * useful / slots, 100 * (ln.sum() - useful) / slots, 100 * (slots - ln.sum()) / slots))
print("   histogram of wave iterations (0,40,60,80,100,120,140,160,180,200,250):", np.histogram(it, bins=[0, 40, 60, 80, 100, 120, 140, 160, 180, 200, 250, 400])[0].tolist())
PY
done
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -2 $O/bench_n1.err
python - $O/bench_n1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("frac %.3f  kernel_avg_ms %.4f  ms_per_step %.4f" % (d["roofline"]["frac"], d["roofline"]["kernel_avg_ms"], d["ms_per_step"]))
for k, v in d.get("secondary", {}).items():
    if isinstance(v, dict):
        print(k, {a: (round(b, 5) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "kernel_avg_ms", "burst_ms_after_idle", "launches_timed", "seconds", "error", "success_rate")},
              "frac=%.3f" % v["roofline"]["frac"] if "roofline" in v else "")
    else:
        print(k, v)
PY

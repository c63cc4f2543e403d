#!/bin/bash
# Round 4, visit d: the IK scheduler knobs under SUSTAINED timing (scripts/ik_ab.py), the per-wave occupancy counters of the shipped schedule, and
# the bench line with the sustained secondary legs.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4d}
mkdir -p $O
cd $R
timeout 600 python scripts/ik_ab.py 100000 2 > $O/ik_ab.jsonl 2> $O/ik_ab.err; cat $O/ik_ab.jsonl | cut -c1-220; tail -2 $O/ik_ab.err
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; tail -3 $O/bench_n1.err
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

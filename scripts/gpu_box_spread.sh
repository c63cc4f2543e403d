#!/bin/bash
# One lease = one box of the pool: the headline bench line (events) and the rocprofv3 --kernel-trace --stats average of the same command, plus the
# sustained IK config-3 and RNE 1e7 times -- run on several leases to record the box-to-box spread (profiles/r04_box_spread.jsonl).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-box}
mkdir -p $O
cd $R
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu > $O/bench.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 40 --warmup 3 --no-cpu --no-secondary > $O/prof.log 2>&1
python - $O <<'PY'
import json, sys, csv, glob, os, socket
O = sys.argv[1]
d = json.loads(open(os.path.join(O, "bench.json")).read().strip().splitlines()[-1])
row = {"events_kernel_avg_ms": d["roofline"]["kernel_avg_ms"], "frac_events": d["roofline"]["frac"], "ms_per_step": d["ms_per_step"]}
for f in glob.glob(os.path.join(O, "prof", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_kin_reg" in r["Name"]:
            row["rocprof_avg_us"] = float(r["AverageNs"]) / 1e3; row["rocprof_calls"] = int(r["Calls"]); row["frac_rocprof"] = 520e6 / (float(r["AverageNs"]) * 1e-9) / 8e12
        if "k_stream_probe" in r["Name"]:
            row["stream_probe_avg_us"] = float(r["AverageNs"]) / 1e3
s = d.get("secondary", {})
for k in ("ik_config3", "rne_config4_1e7", "fleet_config5"):
    if isinstance(s.get(k), dict): row[k + "_ms"] = s[k].get("kernel_avg_ms")
print(json.dumps(row))
open(os.path.join(O, "row.json"), "w").write(json.dumps(row) + "\n")
PY
rm -rf $O/prof

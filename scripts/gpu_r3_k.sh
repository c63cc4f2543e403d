#!/bin/bash
# Round 3, visit k: what a scheduling pass of k_ik costs against an LM iteration (SQ_INSTS_VALU at two pass periods, with the per-wave
# iteration / pass counters of the same launches), and the effective clock of the fp64-bound kernels (GRBM_GUI_ACTIVE / wall time).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3k
mkdir -p $O
cd /tmp
for m in 3 7; do
  rm -f /tmp/ikstats.jsonl
  RTBHIP_IK_STATS=/tmp/ikstats.jsonl timeout 300 python $R/bench_extra.py --what ik --no-cpu --steps 3 --tune ik_pass_mask=$m > $O/ik_mask$m.jsonl 2> /dev/null
  python - /tmp/ikstats.jsonl $m <<'PY'
import json, sys, numpy as np
for i, l in enumerate(open(sys.argv[1])):
    d = json.loads(l); a = np.array(d["per_wave"], dtype=np.int64)
    print("mask", sys.argv[2], "launch", i, "items", d["items"], "grid", d["grid"], "flat", d["flat_chunks"], "wave_iters", int(a[:, 0].sum()), "passes", int(a[:, 1].sum()), "lane_iters", int(a[:, 2].sum()))
PY
  python -c "
import json
for l in open('$O/ik_mask$m.jsonl'):
    d=json.loads(l); print('mask $m', d['metric'][:70], 'avg %.4f min %.4f' % (d['kernel_avg_ms'], d['kernel_min_ms']))"
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_ik_mask$m -o pmc -- python $R/bench_extra.py --what ik --no-cpu --steps 3 --tune ik_pass_mask=$m > $O/pmc_ik_mask$m.log 2>&1 || echo "pmc ik mask $m failed"
done
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_clk_rne -o pmc -- python $R/bench_extra.py --what rne,dyn --no-cpu --steps 5 > $O/pmc_clk_rne.log 2>&1 || echo "pmc clk rne failed"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_clk_head -o pmc -- python $R/bench.py --steps 10 --warmup 2 --no-cpu > $O/pmc_clk_head.log 2>&1 || echo "pmc clk head failed"
python - $O <<'PY'
import csv, sys, collections, glob, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*"))):
    if not os.path.isdir(d): continue
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = collections.OrderedDict()
    for f in cc:
        for r in csv.DictReader(open(f)):
            if "rtbhip" not in r["Kernel_Name"]: continue
            k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-34:])
            rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    seen = collections.Counter()
    for (did, name), c in rows.items():
        seen[name] += 1
        if seen[name] > 12: continue
        ns = dur.get(did, 0)
        clk = c.get("GRBM_GUI_ACTIVE", 0) / ns if ns else 0
        print(os.path.basename(d), did, name, "ns", ns, " ".join("%s=%.6g" % kv for kv in sorted(c.items())), "GHz~%.3f" % clk)
PY
find $O -name "*kernel_trace.csv" -size +4M -delete; find $O -name "*counter_collection.csv" -size +4M -delete

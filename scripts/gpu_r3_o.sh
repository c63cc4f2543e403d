#!/bin/bash
# Round 3, visit o: the tree dynamics terms (k_tree_dyn) -- bench lines and VALU instructions per wave.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3o
mkdir -p $O
cd /tmp
timeout 300 python $R/bench_extra.py --what tree --no-cpu --steps 8 2>/dev/null | cut -c1-260
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_tree -o pmc -- python $R/bench_extra.py --what tree --no-cpu --steps 4 > $O/pmc_tree.log 2>&1 || echo "pmc failed"
python - $O <<'PY'
import csv, sys, collections, glob, os
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_tree", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "rtbhip" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-30:], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for k in ("VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size", "Scratch_Size"):
                if k in r: agg[(r["Kernel_Name"].split("(")[0][-30:], k)] = [float(r[k])]
for k, v in sorted(agg.items()): print(k, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
PY

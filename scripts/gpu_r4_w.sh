#!/bin/bash
# Round 4, visit w: the at-rest instantiation of k_tree_rne (Dynamics.gravload / itorque of URDF robots: rtbhip_tree_rne with qd = NULL) --
# tree parity tests, the tree bench lines, VALU instructions per wave.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4w}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_erobot_dynamics.py tests/test_erobot_rne.py tests/test_01_urdf_fleet.py -q -m gpu 2>&1 | tail -3 | tee $O/pytest_tree.log
cd /tmp
timeout 300 python $R/bench_extra.py --what tree --no-cpu --steps 8 2>/dev/null | tee $O/bench_tree.jsonl | cut -c1-330
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_tree -o pmc -- python $R/bench_extra.py --what tree --no-cpu --steps 4 > $O/pmc_tree.log 2>&1 || echo "pmc failed"
python - $O <<'PY' | tee $O/pmc_tree_summary.txt
import csv, sys, collections, glob, os
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_tree", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "rtbhip" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-34:], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for k in ("VGPR_Count", "Scratch_Size"):
                if k in r: agg[(r["Kernel_Name"].split("(")[0][-34:], k)] = [float(r[k])]
for k, v in sorted(agg.items()): print(k, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
PY
rm -rf $O/pmc_tree

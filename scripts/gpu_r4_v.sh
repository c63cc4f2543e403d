#!/bin/bash
# Round 4, visit v: Dynamics.coriolis of DH arms as one two-field (bilinear) pass per column (the product) against the polar form over full passes
# (variant dyn_polar) -- parity tests, interleaved sustained timing, then VALU instructions per wave of the dyn kernels as shipped.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4v}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_dynamics_terms.py tests/test_erobot_dynamics.py -q -m gpu 2>&1 | tail -3 | tee $O/pytest_dyn.log
V=$R/robotics-toolbox-python_amd/lib/variants
for round in 1 2; do
  for lib in "" $V/dyn_polar.so; do
    RTBHIP_LIB=$lib DYN_AB_TAG=$(basename ${lib:-product}) timeout 300 python scripts/dyn_ab.py 2>/dev/null | tee -a $O/dyn_ab.jsonl
  done
done
cd /tmp
timeout 300 python $R/bench_extra.py --what dyn --no-cpu --steps 8 2>/dev/null | tee $O/bench_dyn.jsonl | cut -c1-260
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_dyn -o pmc -- python $R/bench_extra.py --what dyn --no-cpu --steps 4 > $O/pmc_dyn.log 2>&1 || echo "pmc failed"
python - $O <<'PY' | tee $O/pmc_dyn_summary.txt
import csv, sys, collections, glob, os
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "pmc_dyn", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "rtbhip" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for k in ("VGPR_Count", "Scratch_Size"):
                if k in r: agg[(r["Kernel_Name"].split("(")[0][-40:], k)] = [float(r[k])]
for k, v in sorted(agg.items()): print(k, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))
PY
rm -rf $O/pmc_dyn

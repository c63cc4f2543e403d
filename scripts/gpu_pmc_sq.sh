#!/bin/bash
# SQ counter passes for the compute-bound kernels (rne, ik): instruction mix and stall composition
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
WHAT=${1:-rne}
cd /tmp
i=0
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_${WHAT}_$i -o pmc -- python $R/bench_extra.py --what $WHAT --no-cpu --steps 4 ${EXTRA_ARGS} > $R/gpurun_out/pmc_sq_${WHAT}_$i.log 2>&1 || echo "pmc pass $i failed"
done
cd $R
for d in gpurun_out/pmc_sq_${WHAT}_*/; do python - $d/pmc_counter_collection.csv <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'rtbhip' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[0][-30:], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()): print(k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
done

#!/bin/bash
# second GPU visit: parity tests, bench A/B (register-resident vs run-time-n tile), rocprof stats + PMC
set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
for t in "reg=0" ; do
  python bench.py --steps 50 --warmup 5 --no-cpu --tune $t > gpurun_out/bench_$t.json 2> gpurun_out/bench_$t.err
  echo "$t: $(cat gpurun_out/bench_$t.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_avg_ms"], d["roofline"]["frac"])')"
done
for n in 100000 4000000 16000000; do
  python bench.py --steps 30 --warmup 3 --no-cpu --n $n > gpurun_out/bench_n$n.json 2> gpurun_out/bench_n$n.err
  echo "n=$n: $(cat gpurun_out/bench_n$n.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_avg_ms"], d["roofline"]["frac"])')"
done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 -L > $R/gpurun_out/counters_list.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/pmc_$tag.log 2>&1 || echo "pmc $c failed"
done
cd $R
find gpurun_out/prof_stats -name "*.csv" | head; f=$(find gpurun_out/prof_stats -name "*kernel_stats*.csv" | head -1); head -6 "$f"
for d in gpurun_out/pmc_*/; do f=$(find $d -name "*counter_collection*.csv" | head -1); echo $f; python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_kin' in r.get('Kernel_Name',''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, 'n=%d'%len(v), 'mean=%.6g'%(sum(v)/len(v)))
PY
done

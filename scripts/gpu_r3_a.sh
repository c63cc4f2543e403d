#!/bin/bash
# Round 3, visit a: the whole -m gpu suite (PoE row, the reference's own classes on the shim, RCCL world-size-1 rehearsal,
# frne cross-pin), smoke, headline bench, bench.py --gather under torchrun, secondary legs, kernel stats, VALU-instruction counts
# of the fp64-bound secondary kernels (for their fp64-valu roofline objects).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-600 $O/bench_n1.json
RTBHIP_BENCH_ARGV='["--gpus","1","--steps","30","--warmup","5","--no-cpu","--gather"]' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py > $O/bench_n1_nccl_world1.json 2> $O/bench_nccl.err; cut -c1-400 $O/bench_n1_nccl_world1.json; tail -3 $O/bench_nccl.err
timeout 900 python bench_extra.py > $O/bench_extra.jsonl 2> $O/bench_extra.err; cut -c1-230 $O/bench_extra.jsonl; tail -2 $O/bench_extra.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $O/prof.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_extra -o extra -- python $R/bench_extra.py --no-cpu --steps 6 > $O/prof_extra.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq_sec -o pmc -- python $R/bench_extra.py --what rne,dyn,tree,kin,poe --no-cpu --steps 4 > $O/pmc_sq_sec.log 2>&1 || echo "pmc sq failed"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq_ik -o pmc -- python $R/bench_extra.py --what ik --no-cpu --steps 4 > $O/pmc_sq_ik.log 2>&1 || echo "pmc sq ik failed"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_kin_$c -o pmc -- python $R/bench_extra.py --what kin --no-cpu --steps 3 > $O/pmc_kin_$c.log 2>&1 || echo "pmc kin $c failed"
done
cd $R
find $O/prof $O/prof_extra -name "*kernel_stats*.csv" | while read f; do echo $f; cut -c1-170 "$f" | head -40; done
python - $O <<'PY'
import csv, sys, collections, glob, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rtbhip' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('(')[0][-44:], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()): print(os.path.basename(d), k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete

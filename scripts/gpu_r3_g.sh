#!/bin/bash
# Round 3, visit g (state of the round): the whole -m gpu suite, smoke, headline bench (1 rank; 1 rank with a forced RCCL group + gather;
# 2 ranks sharing the GPU through gloo), every secondary leg, rocprofv3 kernel stats, HBM-traffic PMC passes (headline, rne, partial3),
# SQ counters of k_rne.  Every step has its own timeout.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r3g}
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rf --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -8
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-700 $O/bench_n1.json
RTBHIP_BENCH_ARGV='["--gpus","1","--steps","30","--warmup","5","--no-cpu","--gather"]' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py > $O/bench_n1_nccl_world1.json 2> $O/bench_nccl.err; cut -c1-200 $O/bench_n1_nccl_world1.json
RTBHIP_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu > $O/bench_n2_gloo_shared_gpu.json 2> $O/bench_n2.err; cut -c1-200 $O/bench_n2_gloo_shared_gpu.json
timeout 900 python bench_extra.py > $O/bench_extra.jsonl 2> $O/bench_extra.err; cut -c1-200 $O/bench_extra.jsonl; tail -2 $O/bench_extra.err
RTBHIP_BENCH_BACKEND=gloo timeout 300 python bench_extra.py --gpus 2 --no-cpu > $O/bench_extra_n2_gloo_shared_gpu.jsonl 2> $O/bench_extra_n2.err; cut -c1-160 $O/bench_extra_n2_gloo_shared_gpu.jsonl
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $O/prof.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_extra -o extra -- python $R/bench_extra.py --no-cpu --steps 6 > $O/prof_extra.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_rne_$c -o pmc -- python $R/bench_extra.py --what rne --no-cpu --steps 5 > $O/pmc_rne_$c.log 2>&1 || echo "pmc rne $c failed"
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_kin_$c -o pmc -- python $R/bench_extra.py --what kin,poe --no-cpu --steps 3 > $O/pmc_kin_$c.log 2>&1 || echo "pmc kin $c failed"
done
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq_rne -o pmc -- python $R/bench_extra.py --what rne --no-cpu --steps 4 > $O/pmc_sq_rne.log 2>&1 || echo "pmc sq failed"
cd $R
find $O/prof $O/prof_extra -name "*kernel_stats*.csv" | while read f; do echo $f; cut -c1-170 "$f" | head -16; done
python - $O <<'PY'
import csv, sys, collections, glob, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rtbhip' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()): print(os.path.basename(d), k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete

#!/bin/bash
# GPU visit d: parity, headline bench, RNE A/B after the register-pressure rewrite, rocprof stats, PMC + probe calibration
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
timeout 600 python bench_extra.py --what rne,fleet > gpurun_out/bench_extra_d.jsonl 2> gpurun_out/bench_extra_d.err; cat gpurun_out/bench_extra_d.jsonl; tail -3 gpurun_out/bench_extra_d.err
timeout 300 python bench_extra.py --what rne --no-cpu --n-rne 10000000 > gpurun_out/bench_rne_1e7.jsonl 2>&1; cat gpurun_out/bench_rne_1e7.jsonl
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/prof_d.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_d_rne -o rne -- python $R/bench_extra.py --what rne --no-cpu > $R/gpurun_out/prof_d_rne.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_d_$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/pmc_d_$c.log 2>&1 || echo "pmc $c failed"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_d_probe_$c -o pmc -- $R/scripts/roofline_probe.bin > $R/gpurun_out/pmc_d_probe_$c.log 2>&1 || echo "pmc probe $c failed"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_d_rne_$c -o pmc -- python $R/bench_extra.py --what rne --no-cpu --steps 5 > $R/gpurun_out/pmc_d_rne_$c.log 2>&1 || echo "pmc rne $c failed"
done
cd $R
find gpurun_out/prof_d gpurun_out/prof_d_rne -name "*kernel_stats*.csv" | while read f; do echo $f; head -4 "$f"; done

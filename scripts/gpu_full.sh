#!/bin/bash
# Full GPU visit: parity suite, headline bench, every secondary bench leg, rocprofv3 kernel stats, PMC traffic passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cut -c1-600 gpurun_out/bench_default.json
timeout 900 python bench_extra.py > gpurun_out/bench_extra_all.jsonl 2> gpurun_out/bench_extra_all.err; cut -c1-200 gpurun_out/bench_extra_all.jsonl; tail -2 gpurun_out/bench_extra_all.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $R/gpurun_out/prof_${TAG}.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_extra -o extra -- python $R/bench_extra.py --no-cpu --steps 6 > $R/gpurun_out/prof_${TAG}_extra.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1 || echo "pmc $c failed"
done
cd $R
find gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_extra -name "*kernel_stats*.csv" | while read f; do echo $f; cut -c1-160 "$f" | head -16; done

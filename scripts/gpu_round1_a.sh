#!/bin/bash
# first GPU visit: parity tests, bench A/B of store path and launch geometry, rocprof kernel stats
set -x
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6 > gpurun_out/rocminfo.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
cat gpurun_out/bench_default.json
for t in "coalesced=0" "tiles_per_wave=2" "tiles_per_wave=4" "tiles_per_wave=10"; do
  python bench.py --steps 50 --warmup 5 --no-cpu --tune $t > gpurun_out/bench_$t.json 2> gpurun_out/bench_$t.err
  echo "$t: $(cat gpurun_out/bench_$t.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_avg_ms"], d["roofline"]["frac"])')"
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_bench -name "*stats*" | head
f=$(find gpurun_out/prof_bench -name "*kernel_stats*.csv" | head -1); head -8 "$f"

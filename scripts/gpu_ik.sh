#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "ik" 2>&1 | tail -8
timeout 300 python bench_extra.py --what ik > gpurun_out/bench_ik.jsonl 2> gpurun_out/bench_ik.err; cat gpurun_out/bench_ik.jsonl; tail -3 gpurun_out/bench_ik.err
for w in 4 12 16; do echo "waves_per_cu=$w"; timeout 300 python bench_extra.py --what ik --no-cpu --tune ik_waves_per_cu=$w 2>&1 | tail -1 | cut -c1-400; done
echo "1e6 targets"; timeout 300 python bench_extra.py --what ik --no-cpu --n-ik 1000000 2>&1 | tail -1 | cut -c1-400

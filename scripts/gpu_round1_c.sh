#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -25 gpurun_out/pytest_gpu.log
python bench.py --steps 50 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
timeout 600 python bench_extra.py > gpurun_out/bench_extra.jsonl 2> gpurun_out/bench_extra.err; cat gpurun_out/bench_extra.jsonl; tail -3 gpurun_out/bench_extra.err

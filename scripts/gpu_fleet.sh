#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python bench_extra.py --what fleet > gpurun_out/bench_fleet.jsonl 2> gpurun_out/bench_fleet.err; cat gpurun_out/bench_fleet.jsonl; tail -5 gpurun_out/bench_fleet.err

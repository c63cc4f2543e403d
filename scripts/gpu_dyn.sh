#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 600 python bench_extra.py --what dyn > gpurun_out/bench_dyn.jsonl 2> gpurun_out/bench_dyn.err; cat gpurun_out/bench_dyn.jsonl; tail -5 gpurun_out/bench_dyn.err

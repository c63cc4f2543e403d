#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_hip_graph.py -m gpu -x -q > gpurun_out/pytest_graph.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_graph.log
tail -15 gpurun_out/pytest_graph.log
timeout 300 python bench_extra.py --what graph --no-cpu 2>&1 | tail -3

#!/bin/bash
# gpu_one.sh <pytest args...> : run a subset of the GPU tests
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest "$@" -m gpu -x -q > gpurun_out/pytest_one.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_one.log
grep -v Warning gpurun_out/pytest_one.log | tail -25

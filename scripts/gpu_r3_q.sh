#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3q
timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 600 > gpurun_out/r3q/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3q/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" gpurun_out/r3q/pytest_gpu.log | tail -15

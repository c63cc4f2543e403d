#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fkine_all.py -m gpu -x -q 2>&1 | grep -v Warning | tail -3
timeout 600 python bench_extra.py --what kin --no-cpu 2>/dev/null | grep fkine_all | cut -c1-330

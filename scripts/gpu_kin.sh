#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python bench_extra.py --what kin --no-cpu 2>/dev/null | tee gpurun_out/bench_kin.jsonl | cut -c1-250

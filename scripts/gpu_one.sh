#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 200 python bench_extra.py --what ik --steps 16 2>/dev/null | grep '^{' > gpurun_out/r2n/bench_ik.jsonl; cut -c1-330 gpurun_out/r2n/bench_ik.jsonl

#!/bin/bash
# Round 2, visit l: multi-rank legs of bench_extra (now with the IK config) on two ranks sharing the GPU; dist tests
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 600 python -m pytest tests/test_dist_gpu.py -m gpu -q --timeout 300 --tb=short 2>&1 | grep -v "Warning\|^  \|^$" | tail -6
RTBHIP_BENCH_BACKEND=gloo timeout 300 python bench_extra.py --gpus 2 --no-cpu 2>/dev/null | grep '^{' > gpurun_out/r2l/bench_extra_n2_gloo_shared_gpu.jsonl; cut -c1-260 gpurun_out/r2l/bench_extra_n2_gloo_shared_gpu.jsonl

#!/bin/bash
# multi-rank code path of bench.py on a 1-GPU box (2 ranks share the GPU, gloo collectives): a flow check, not a measurement
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
RTBHIP_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 2>&1 | tail -4 | cut -c1-700

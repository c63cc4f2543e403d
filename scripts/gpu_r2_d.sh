#!/bin/bash
# Round 2, visit d: full GPU suite after the joint-count widening (IK 16, dynamics terms 16, diff kinematics 16, tree 24 groups,
# null-space 6..12), loop-time trace of the rne bench leg, secondary bench lines
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2d
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR" $O/pytest_gpu.log | tail -12
RTBHIP_BENCH_TRACE=1 python bench_extra.py --what rne --no-cpu --steps 30 2>&1 | cut -c1-700
RTBHIP_BENCH_TRACE=1 python bench_extra.py --what rne --no-cpu --steps 30 --warmup 40 2>&1 | cut -c1-400
timeout 900 python bench_extra.py > $O/bench_extra_all.jsonl 2> $O/bench_extra_all.err; cut -c1-230 $O/bench_extra_all.jsonl; tail -2 $O/bench_extra_all.err

#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_diff_kinematics.py tests/test_jacob_reference_style.py -m gpu -x -q > gpurun_out/pytest_partial.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_partial.log
tail -15 gpurun_out/pytest_partial.log
timeout 600 python bench_extra.py --what kin --no-cpu > gpurun_out/bench_kin.jsonl 2> gpurun_out/bench_kin.err; cut -c1-330 gpurun_out/bench_kin.jsonl; tail -3 gpurun_out/bench_kin.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_partial -o kin -- python $R/bench_extra.py --what kin --no-cpu --steps 6 > $R/gpurun_out/prof_partial.log 2>&1
cd $R
find gpurun_out/prof_partial -name "*kernel_stats*.csv" | while read f; do cut -c1-200 "$f" | head -12; done

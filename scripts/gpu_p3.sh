#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_diff_kinematics.py -m gpu -x -q -k partial 2>&1 | grep -v Warning | tail -2
./scripts/gpu_p4.sh
cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o k -- python $GRAFT_REPO_ROOT/bench_extra.py --what kin --no-cpu --steps 4 > /dev/null 2>&1; grep k_partial /tmp/pp/k_kernel_stats.csv | cut -c1-140

#!/bin/bash
# Round 3, visit z: dynamics terms, the round-2 tree against the present one (coriolis after the one-copy restructuring; accel with and without
# the p* = 0 shortcut), interleaved on one box; then the dynamics tests.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # dir label args
  d=$1; l=$2; shift 2
  (cd $d && timeout 300 python bench_extra.py --what dyn --no-cpu --steps 12 "$@" 2>/dev/null) | python -c "
import sys,json
print('$l', ' | '.join('%s %.4f (min %.4f)' % (json.loads(x)['metric'].split()[-1][:-1][:8], json.loads(x)['kernel_avg_ms'], json.loads(x)['kernel_min_ms']) for x in sys.stdin if x.startswith('{')))"
}
for rep in 1 2 3; do
  run $R/r2cmp "r2          "
  run $R       "r3          "

done
cd $R && timeout 900 python -m pytest tests/test_dynamics_terms.py tests/test_erobot_dynamics.py -m gpu -q -x 2>&1 | tail -2

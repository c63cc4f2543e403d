#!/bin/bash
# Round 3, visit s: the unit-weight LM step against the weighted one on the same mask of ones, interleaved.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do
for u in 1 0; do
timeout 300 python bench_extra.py --what ik --no-cpu --steps 8 --tune ik_unit_we=$u 2>/dev/null | python -c "
import sys,json
print('unit_we=$u', ' | '.join('%.4f (min %.4f)' % (json.loads(l)['kernel_avg_ms'], json.loads(l)['kernel_min_ms']) for l in sys.stdin))"
done
done

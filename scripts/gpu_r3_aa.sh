#!/bin/bash
# Round 3, visit aa: partial_fkine0 order 3 with the Hessians formed inside k_partial3 against the two-launch form (interleaved); the partial and
# coriolis tile-invariance tests.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "partial or tile_mates" 2>&1 | tail -3
for rep in 1 2 3; do
for f in 1 0; do
timeout 300 python bench_extra.py --what kin --no-cpu --steps 8 --tune partial3_fused=$f 2>/dev/null | grep partial_fkine0 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('fused=$f', {k: round(v, 4) for k, v in d.items() if 'ms' in k and isinstance(v, float)}, 'frac', round(d['roofline']['frac'], 3))"
done
done

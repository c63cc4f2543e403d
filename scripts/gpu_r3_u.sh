#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT

timeout 900 python bench_extra.py --no-cpu --steps 6 2>/dev/null | grep "mixed fleet" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['metric'][:60], {k: round(v,4) for k,v in d.items() if 'ms' in k and isinstance(v, float)})"

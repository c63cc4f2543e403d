#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2; do for m in 1 3 7; do echo -n "pass_mask $m: "; timeout 120 python bench_extra.py --what ik --no-cpu --steps 16 --tune ik_pass_mask=$m 2>/dev/null | python -c '
import json,sys
print(" | ".join("n=%d avg %.3f min %.3f" % (d["n"], d["kernel_avg_ms"], d["kernel_min_ms"]) for d in map(json.loads, sys.stdin)))'; done; done

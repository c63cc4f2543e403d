#!/bin/bash
export TMPDIR=/tmp
for pm in 1 3 7; do
  echo "pass_mask=$pm"; timeout 300 python bench_extra.py --what ik --no-cpu --tune ik_pass_mask=$pm 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print("   %-60s avg %.4f ms min %.4f" % (d["metric"][:60], d["kernel_avg_ms"], d["kernel_min_ms"]))'
done

#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" $R/robotics-toolbox-python_amd/lib/variants/base.so; do
  echo "== ${v:-new}"
  RTBHIP_LIB=$v timeout 600 python bench_extra.py --what rne,dyn --no-cpu 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print("%-50s avg %.4f ms min %.4f ms" % (d["metric"][:50], d["kernel_avg_ms"], d["kernel_min_ms"]))'
done; done

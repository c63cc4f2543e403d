#!/bin/bash
# A/B: default library vs every variant under lib/variants on the rne leg (interleaved, 3 rounds)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "" robotics-toolbox-python_amd/lib/variants/*.so; do
    n=${v:-default}
    RTBHIP_LIB=${v:+$GRAFT_REPO_ROOT/$v} timeout 120 python bench_extra.py --what rne --no-cpu --steps 40 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print("'$(basename $n .so)'", "avg %.4f min %.4f" % (d["kernel_avg_ms"], d["kernel_min_ms"]))'
  done
done

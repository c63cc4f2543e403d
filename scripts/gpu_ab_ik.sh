#!/bin/bash
# A/B: default library vs every variant under lib/variants on the IK leg (interleaved, 3 rounds)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "" robotics-toolbox-python_amd/lib/variants/*.so; do
    n=${v:-default}
    RTBHIP_LIB=${v:+$GRAFT_REPO_ROOT/$v} timeout 120 python bench_extra.py --what ik --no-cpu --steps 16 2>/dev/null | python -c '
import json,sys
out=[]
for l in sys.stdin:
    d=json.loads(l); out.append("n=%d avg %.3f min %.3f" % (d["n"], d["kernel_avg_ms"], d["kernel_min_ms"]))
print("'$(basename $n .so)'", " | ".join(out))'
  done
done

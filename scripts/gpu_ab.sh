#!/bin/bash
# A/B: default library vs every variant under lib/variants on the headline bench (interleaved, 3 rounds)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "" robotics-toolbox-python_amd/lib/variants/*.so; do
    n=${v:-default}
    RTBHIP_LIB=${v:+$GRAFT_REPO_ROOT/$v} python bench.py --steps 100 --warmup 10 --no-cpu 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("'$(basename $n .so)'", "%.4g cfg/s step %.4f ms kernel avg %.4f min %.4f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_ms"], d["roofline"]["kernel_min_ms"], d["roofline"]["frac"]))'
  done
done

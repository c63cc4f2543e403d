#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
for v in "" $R/robotics-toolbox-python_amd/lib/variants/*.so; do
  RTBHIP_LIB=$v python bench.py --steps 200 --warmup 20 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('${v:-eighths}'[-12:], 'kernel avg %.4f ms min %.4f  frac %.3f' % (d['roofline']['kernel_avg_ms'], d['roofline']['kernel_min_ms'], d['roofline']['frac']))"
done; done

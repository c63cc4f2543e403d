#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q -k "rne" 2>&1 | tail -3
for v in "" $R/robotics-toolbox-python_amd/lib/variants/rne_*.so; do
  echo "== $(basename $v)"
  RTBHIP_LIB=$v python bench_extra.py --what rne --no-cpu 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4g triples/s avg %.4f ms min %.4f ms" % (d["value"], d["kernel_avg_ms"], d["kernel_min_ms"]))'
  RTBHIP_LIB=$v python bench_extra.py --what rne --no-cpu --n-rne 10000000 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("1e7: %.4g triples/s avg %.4f ms min %.4f ms" % (d["value"], d["kernel_avg_ms"], d["kernel_min_ms"]))'
done

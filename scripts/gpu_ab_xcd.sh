#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -1
for rep in 1 2 3; do
for v in "" $R/robotics-toolbox-python_amd/lib/variants/noremap.so; do
  echo "== ${v:-remap} (rep $rep)"
  RTBHIP_LIB=$v python bench.py --steps 200 --warmup 20 --no-cpu | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   headline kernel avg %.4f ms min %.4f  frac %.3f' % (d['roofline']['kernel_avg_ms'], d['roofline']['kernel_min_ms'], d['roofline']['frac']))"
  RTBHIP_LIB=$v timeout 600 python bench_extra.py --what rne,kin,fleet,dyn,tree --no-cpu 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if "kernel_avg_ms" in d: print("   %-58s avg %.4f" % (d["metric"][:58], d["kernel_avg_ms"]))
    elif "call_avg_ms" in d: print("   %-58s call %.4f" % (d["metric"][:58], d["call_avg_ms"]))'
done; done

#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "rne or dyn or inertia or coriolis or accel or G9 or graph" 2>&1 | grep -v Warning | tail -3
timeout 600 python bench_extra.py --what rne,dyn --no-cpu 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print("%-50s avg %.4f ms min %.4f ms" % (d["metric"][:50], d["kernel_avg_ms"], d["kernel_min_ms"]))'

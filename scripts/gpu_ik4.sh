#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "ik" 2>&1 | grep -v Warning | tail -3
timeout 600 python bench_extra.py --what ik --no-cpu 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l); print("%-70s avg %.4f ms min %.4f ms success %.4f" % (d["metric"][:70], d["kernel_avg_ms"], d["kernel_min_ms"], d["success_rate"]))'

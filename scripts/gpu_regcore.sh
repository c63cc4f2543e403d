#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -3
for i in 1 2; do python bench.py --steps 100 --warmup 10 --no-cpu | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("headline %.4g cfg/s  kernel avg %.4f ms min %.4f" % (d["value"], d["roofline"]["kernel_avg_ms"], d["roofline"]["kernel_min_ms"]))'; done
timeout 600 python bench_extra.py --what ik,kin --no-cpu 2>/dev/null | python -c '
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if "kernel_avg_ms" in d: print("%-70s avg %.4f ms min %.4f ms" % (d["metric"][:70], d["kernel_avg_ms"], d["kernel_min_ms"]))'

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "ik" 2>&1 | tail -3
for i in 1 2; do for n in 100000 1000000; do timeout 300 python bench_extra.py --what ik --no-cpu --n-ik $n 2>&1 | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("  n=%d avg %.3f ms min %.3f ms  %.4g solves/s" % (d["n"], d["kernel_avg_ms"], d["kernel_min_ms"], d["value"]))'; done; done

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "rne or dynamics or G9" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench_extra.py --what rne --no-cpu 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("rne 1.25e6: avg %.4f ms min %.4f ms  %.4g triples/s" % (d["kernel_avg_ms"], d["kernel_min_ms"], d["value"]))'; done
timeout 300 python bench_extra.py --what rne --no-cpu --n-rne 10000000 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("rne 1e7: avg %.4f ms min %.4f ms  %.4g triples/s" % (d["kernel_avg_ms"], d["kernel_min_ms"], d["value"]))'

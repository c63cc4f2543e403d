#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3v
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu > gpurun_out/r3v/bench_n1.json 2> gpurun_out/r3v/bench.err; python -c "
import json; d=json.loads(open('gpurun_out/r3v/bench_n1.json').read().strip().split('\n')[-1]); r=d['roofline']; print(d['ms_per_step'], r['frac'], r.get('stream_probe'), r.get('frac_of_stream_probe'))"


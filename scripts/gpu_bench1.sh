#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python -c "
import json; d=json.load(open('gpurun_out/bench_default.json')); print(d['value'], d['roofline']['frac'], json.dumps(d['cpu_baseline'])[:900])"; tail -2 gpurun_out/bench_default.err; nproc

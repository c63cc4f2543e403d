#!/bin/bash
# Round 2, visit a: full GPU parity suite (incl. the 2-rank tests), headline bench at N=1 and as 2 self-spawned ranks
# (gloo hook, one GPU shared), the sharded config-4/5 legs, the secondary bench legs.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 200 --warmup 20 > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-1500 $O/bench_n1.json; tail -3 $O/bench_n1.err
python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_n1_driver_args.json 2>> $O/bench_n1.err; cut -c1-400 $O/bench_n1_driver_args.json
RTBHIP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err; cut -c1-900 $O/bench_n2_gloo.json; tail -3 $O/bench_n2_gloo.err
RTBHIP_BENCH_BACKEND=gloo timeout 600 python bench_extra.py --gpus 2 --what rne,fleet --no-cpu --steps 6 > $O/extra_n2_gloo.jsonl 2> $O/extra_n2_gloo.err; cut -c1-500 $O/extra_n2_gloo.jsonl; tail -3 $O/extra_n2_gloo.err
timeout 900 python bench_extra.py > $O/bench_extra_all.jsonl 2> $O/bench_extra_all.err; cut -c1-260 $O/bench_extra_all.jsonl; tail -2 $O/bench_extra_all.err

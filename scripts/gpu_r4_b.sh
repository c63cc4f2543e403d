#!/bin/bash
# Round 4, visit b: the gate first -- the whole -m gpu suite WITHOUT -x (a peripheral failure must not hide the rest), smoke, headline bench.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4b}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-900 $O/bench_n1.json

#!/bin/bash
# Round 2, visit g: where the time of an IK call with cross-wave sharing goes (kernel trace of both modes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g
mkdir -p $O
cd /tmp
for sh in 0 1; do
  timeout 150 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_share$sh -o ik -- python $R/bench_extra.py --what ik --no-cpu --steps 12 --tune ik_share=$sh > $O/bench_share$sh.txt 2>&1
  f=$(find $O/prof_share$sh -name "*kernel_stats.csv" | head -1)
  echo "== share $sh: $f"; head -12 "$f" | cut -c1-200
done

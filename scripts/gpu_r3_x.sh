#!/bin/bash
# Round 3, visit x: state of the round (scripts/gpu_r3_g.sh) and, on the same box, the round-2 final tree (r2cmp/ = commit 61a4a33, untracked)
# against the present one: rne, IK, dynamics terms, headline.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
VISIT=r3x bash $R/scripts/gpu_r3_g.sh
O=$R/gpurun_out/r3x
run() { # dir label args...
  d=$1; l=$2; shift 2
  (cd $d && timeout 300 python bench_extra.py "$@" 2>/dev/null) | python -c "
import sys,json
for ln in sys.stdin:
    if not ln.startswith('{'): continue
    d=json.loads(ln); print('$l', d['metric'][:52].ljust(52), 'step %.4f' % d.get('ms_per_step', 0), 'kernel avg %.4f min %.4f' % (d.get('kernel_avg_ms', 0), d.get('kernel_min_ms', 0)))"
}
if [ -d $R/r2cmp ] && [ -z "$SKIP_R2" ]; then
{
for rep in 1 2; do
  run $R/r2cmp "r2 " --what rne --no-cpu --steps 30
  run $R       "r3 " --what rne --no-cpu --steps 30
  run $R/r2cmp "r2 " --what rne --no-cpu --steps 10 --n-rne 10000000
  run $R       "r3 " --what rne --no-cpu --steps 10 --n-rne 10000000
  run $R/r2cmp "r2 " --what ik --no-cpu --steps 8
  run $R       "r3 " --what ik --no-cpu --steps 8
done
run $R/r2cmp "r2 " --what dyn,kin --no-cpu --steps 8
run $R       "r3 " --what dyn,kin --no-cpu --steps 8
for rep in 1 2; do
(cd $R/r2cmp && timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu 2>/dev/null) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('r2  headline', d['ms_per_step'])"
(cd $R && timeout 200 python bench.py --steps 50 --warmup 5 --no-cpu 2>/dev/null) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('r3  headline', d['ms_per_step'], d['roofline'].get('frac_of_stream_probe'))"
done
} > $O/r2_vs_r3.txt 2>&1
cat $O/r2_vs_r3.txt
fi

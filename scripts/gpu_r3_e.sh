#!/bin/bash
# Round 3, visit e: IK at the config-3 size -- waves per CU x schedule x share of the batch started per pass.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for w in 8 6 4; do for f in 1 0; do for pct in 50 100; do
  timeout 300 python bench_extra.py --what ik --no-cpu --steps 12 --tune ik_waves_per_cu=$w --tune ik_flat=$f --tune ik_fresh_pct=$pct 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves_per_cu=$w flat=$f fresh_pct=$pct', 'avg %.4f min %.4f ms' % (d['kernel_avg_ms'], d['kernel_min_ms']), '%.3g' % d['lm_iterations_per_s'])"
done; done; done
for l in "2 4" "2 8" "3 6" "4 6" "6 8" "4 12"; do set -- $l
  timeout 300 python bench_extra.py --what ik --no-cpu --steps 12 --tune ik_flat_l0=$1 --tune ik_flat_len=$2 2>/dev/null | head -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flat l0=$1 len=$2', 'avg %.4f min %.4f ms' % (d['kernel_avg_ms'], d['kernel_min_ms']))"
done

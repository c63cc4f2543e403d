#!/bin/bash
# collect_visit.sh r4n n : copy what a full visit (scripts/gpu_r4_c.sh layout) left under gpurun_out/<visit>/ into profiles/r04_<letter>_* and
# regenerate profiles/r04_pmc.json / r04_pmc_rne.json from its PMC passes.
set -e
V=$1; L=$2; O=gpurun_out/$V; P=profiles
cp $O/pytest_gpu.log $P/r04_${L}_pytest_gpu.log
cp $O/bench_n1.json $P/r04_${L}_bench_n1.json
cp $O/bench_extra.jsonl $P/r04_${L}_bench_extra.jsonl
cp $O/prof/bench_kernel_stats.csv $P/r04_${L}_kernel_stats.csv
cp $O/prof_rne1e7/rne_kernel_stats.csv $P/r04_${L}_rne1e7_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  cp $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) $P/r04_${L}_pmc_$c.csv
  cp $(find $O/pmc_rne_$c -name "*counter_collection.csv" | head -1) $P/r04_${L}_pmc_rne_$c.csv
done
python scripts/pmc_summary.py --kernel k_kin_reg --fetch $P/r04_${L}_pmc_FETCH_SIZE.csv --write $P/r04_${L}_pmc_WRITE_SIZE.csv --algorithmic-bytes 520e6 --out $P/r04_pmc.json > /dev/null
python scripts/pmc_summary.py --kernel k_rne --fetch $P/r04_${L}_pmc_rne_FETCH_SIZE.csv --write $P/r04_${L}_pmc_rne_WRITE_SIZE.csv --algorithmic-bytes 280e6 --out $P/r04_pmc_rne.json > /dev/null
python - $L <<'PY'
import json, sys
for n in ('r04_pmc.json', 'r04_pmc_rne.json'):
    p = 'profiles/' + n; d = json.load(open(p))
    d['visit'] = 'round 4 visit %s (same lease as profiles/r04_%s_bench_n1.json and r04_%s_kernel_stats.csv)' % (sys.argv[1], sys.argv[1], sys.argv[1])
    json.dump(d, open(p, 'w'), indent=1); print(n, d['hbm_bytes_per_launch'], d.get('traffic_over_algorithmic'))
PY
head -3 $P/r04_${L}_kernel_stats.csv | cut -c1-200

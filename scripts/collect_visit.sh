#!/bin/bash
# collect_visit.sh VISIT LETTER [ROUND] : copy what a full visit of scripts/visit.sh (STAGES="tests smoke bench prof [profsec] pmc [sq iksq] extra [fuzz]") left under
# gpurun_out/VISIT/ into profiles/rNN_<letter>_* and regenerate profiles/rNN_pmc.json / rNN_pmc_rne.json from its PMC passes (ROUND default 05).
set -e
V=$1; L=$2; RN=${3:-05}; O=gpurun_out/$V; P=profiles
cp $O/pytest_gpu.log $P/r${RN}_${L}_pytest_gpu.log
cp $O/bench_n1.json $P/r${RN}_${L}_bench_n1.json
[ -f $O/bench_extra.jsonl ] && cp $O/bench_extra.jsonl $P/r${RN}_${L}_bench_extra.jsonl
cp $O/prof/bench_kernel_stats.csv $P/r${RN}_${L}_kernel_stats.csv
[ -f $O/prof_rne1e7/rne_kernel_stats.csv ] && cp $O/prof_rne1e7/rne_kernel_stats.csv $P/r${RN}_${L}_rne1e7_kernel_stats.csv
[ -f $O/secondary_kernel_stats.csv ] && cp $O/secondary_kernel_stats.csv $P/r${RN}_${L}_secondary_kernel_stats.csv
[ -f $O/sq_digest.txt ] && cp $O/sq_digest.txt $P/r${RN}_${L}_sq_digest.txt
[ -f $O/ik_loss_factors.json ] && cp $O/ik_loss_factors.json $P/r${RN}_${L}_ik_loss_factors.json
[ -f $O/ik_sq.json ] && python - $O/ik_sq.json $P/r${RN}_ik_sq.json $V <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
d["visit"] = "%s (gpurun_out/%s/pmc_iksq: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace over scripts/ik_loss_factors.py --launches 4)" % (sys.argv[3], sys.argv[3])
json.dump(d, open(sys.argv[2], "w"), indent=1)
PY
ls $O/fuzz_*.jsonl > /dev/null 2>&1 && for f in $O/fuzz_*.jsonl; do echo "# $(basename $f)"; tail -3 $f; done > $P/r${RN}_${L}_fuzz.txt
for c in FETCH_SIZE WRITE_SIZE; do
  cp $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) $P/r${RN}_${L}_pmc_$c.csv
  cp $(find $O/pmc_rne_$c -name "*counter_collection.csv" | head -1) $P/r${RN}_${L}_pmc_rne_$c.csv
done
python scripts/pmc_summary.py --kernel k_kin_reg --fetch $P/r${RN}_${L}_pmc_FETCH_SIZE.csv --write $P/r${RN}_${L}_pmc_WRITE_SIZE.csv --algorithmic-bytes 520e6 --out $P/r${RN}_pmc.json > /dev/null
python scripts/pmc_summary.py --kernel k_rne --fetch $P/r${RN}_${L}_pmc_rne_FETCH_SIZE.csv --write $P/r${RN}_${L}_pmc_rne_WRITE_SIZE.csv --algorithmic-bytes 280e6 --out $P/r${RN}_pmc_rne.json > /dev/null
python - $L $RN <<'PY'
import json, sys
L, RN = sys.argv[1], sys.argv[2]
for n in ('r%s_pmc.json' % RN, 'r%s_pmc_rne.json' % RN):
    p = 'profiles/' + n; d = json.load(open(p))
    d['visit'] = 'round %d visit %s (same lease as profiles/r%s_%s_bench_n1.json and r%s_%s_kernel_stats.csv)' % (int(RN), L, RN, L, RN, L)
    json.dump(d, open(p, 'w'), indent=1); print(n, d['hbm_bytes_per_launch'], d.get('traffic_over_algorithmic'))
PY
head -3 $P/r${RN}_${L}_kernel_stats.csv | cut -c1-200

#!/bin/bash
# Round 3, visit b: the whole -m gpu suite after the hygiene / ABI batch (device scope, uploads, trim, array pi, k_partial3),
# k_partial3 against the general kernel (time, HBM bytes), bench lines.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3b
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rf --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for t in 1 0 1 0; do
  timeout 300 python bench_extra.py --what kin --no-cpu --steps 20 --tune partial3=$t 2>/dev/null | grep partial | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('partial3=$t', d['call_avg_ms'], d['call_min_ms'], d['roofline']['frac'])"
done
cd /tmp
for t in 1 0; do
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_p${t}_$c -o pmc -- python $R/bench_extra.py --what kin --no-cpu --steps 3 --tune partial3=$t > $O/pmc_p${t}_$c.log 2>&1 || echo "pmc failed"
done
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kin -o kin -- python $R/bench_extra.py --what kin --no-cpu --steps 10 > $O/prof_kin.log 2>&1
cd $R
find $O/prof_kin -name "*kernel_stats*.csv" | while read f; do grep -E "partial|hess_tile|Name" "$f" | cut -c1-200; done
python - $O <<'PY'
import csv, sys, collections, glob, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'partial' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('(')[0][-44:], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()): print(os.path.basename(d), k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete

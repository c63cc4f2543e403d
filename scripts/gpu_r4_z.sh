#!/bin/bash
# Round 4, visit c: ONE lease for everything the roofline is computed from -- the whole -m gpu suite WITHOUT -x, smoke, the bench line (with the
# `secondary` object), rocprofv3 --kernel-trace --stats of the same command, the FETCH_SIZE / WRITE_SIZE passes (headline; rne leg), the k_rne
# probe at the full config-4 size, and the secondary benches.  Results: gpurun_out/$VISIT.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4z}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -12
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-400 $O/bench_n1.json; tail -3 $O/bench_n1.err
python - $O/bench_n1.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("frac %.3f  kernel_avg_ms %.4f  ms_per_step %.4f" % (d["roofline"]["frac"], d["roofline"]["kernel_avg_ms"], d["ms_per_step"]))
    for k, v in d.get("secondary", {}).items():
        if isinstance(v, dict):
            print(k, {a: (round(b, 5) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "kernel_avg_ms", "seconds", "error", "success_rate")},
                  "frac=%.3f" % v["roofline"]["frac"] if "roofline" in v else "", v.get("parity") if not isinstance(v.get("parity"), str) else "")
        else:
            print(k, v)
except Exception as e:
    print("bench line unreadable:", e)
PY
timeout 300 python scripts/rne_1e7_probe.py > $O/rne_1e7_probe.json 2> $O/rne_1e7_probe.err; cut -c1-1200 $O/rne_1e7_probe.json; tail -2 $O/rne_1e7_probe.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 40 --warmup 3 --no-cpu --no-secondary > $O/prof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-secondary > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_rne_$c -o pmc -- python $R/bench_extra.py --what rne --no-cpu --steps 5 > $O/pmc_rne_$c.log 2>&1 || echo "pmc rne $c failed"
done
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_clk_rne -o pmc -- python $R/bench_extra.py --what rne --no-cpu --steps 20 --n-rne 10000000 > $O/pmc_clk_rne.log 2>&1 || echo "pmc clk failed"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rne1e7 -o rne -- python $R/bench_extra.py --what rne --no-cpu --steps 40 --n-rne 10000000 > $O/prof_rne1e7.log 2>&1
cd $R
timeout 900 python bench_extra.py > $O/bench_extra.jsonl 2> $O/bench_extra.err; cut -c1-160 $O/bench_extra.jsonl; tail -2 $O/bench_extra.err
find $O/prof $O/prof_rne1e7 -name "*kernel_stats*.csv" | while read f; do echo $f; cut -c1-170 "$f" | head -8; done
python - $O <<'PY'
import csv, sys, collections, glob, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rtbhip' in r['Kernel_Name']:
                agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(agg.items()): print(os.path.basename(d), k, 'n=%d' % len(v), 'mean=%.6g' % (sum(v) / len(v)))
PY
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
# every size of every kernel family on the device against the oracle, random robots (scripts/gpu_fuzz_*.py)
cd $R
for f in dyn ik kin rne paths fleet; do
  timeout 900 python scripts/gpu_fuzz_$f.py > $O/fuzz_$f.jsonl 2> $O/fuzz_$f.err; echo "fuzz $f rc=$?" >> $O/fuzz_$f.jsonl; tail -2 $O/fuzz_$f.jsonl | cut -c1-300
done

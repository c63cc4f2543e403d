#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the secondary kernels
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcx_$c -o pmc -- python $R/bench_extra.py --what rne,kin,fleet,tree,dyn --no-cpu --steps 3 > $R/gpurun_out/pmcx_$c.log 2>&1 || echo "pmc $c failed"
done
cd $R
python - <<'PY'
import csv, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/pmcx_%s/pmc_counter_collection.csv" % c)):
        if "rtbhip" in r["Kernel_Name"] and r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0].replace("void rtbhip::", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[c] = (max(v), len(v))
for k in sorted(out):
    f = out[k].get("FETCH_SIZE", (0, 0)); w = out[k].get("WRITE_SIZE", (0, 0))
    print("%-42s read %9.1f MB (x2 corrected)  written %9.1f MB   [largest launch of %d]" % (k[:42], f[0] * 1024 * 2 / 1e6, w[0] * 1024 / 1e6, w[1]))
PY

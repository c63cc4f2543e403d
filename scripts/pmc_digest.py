#!/usr/bin/env python3
"""scripts/pmc_digest.py DIR -- mean counter value per (kernel, counter) of every rocprofv3 --pmc pass under DIR/pmc_*."""
import collections
import csv
import glob
import os
import sys
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "rtbhip" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            print(os.path.basename(d), k, "n=%d" % len(v), "mean=%.6g" % (sum(v) / len(v)))

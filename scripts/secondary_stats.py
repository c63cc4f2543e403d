"""scripts/secondary_stats.py <dir with rocprofv3 --kernel-trace csv> <out.csv>
Per (kernel, grid size) durations of one `rocprofv3 --kernel-trace --stats -- python bench.py` run WITH the secondary legs: the stats summary
rocprofv3 writes groups by kernel name only, and config 4 launches the same k_rne at 1e7 triples and at its 1.25e6 share.  Columns:
Name, GridSize (work-items), Calls, AverageNs, MinNs, MaxNs, TotalDurationNs -- what benchsecondary._committed reads back."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(d, out):
    files = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit("no *kernel_trace.csv under " + d)
    acc = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            try:
                t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                grid = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0) * max(1, int(r.get("Grid_Size_Y") or 1)) * max(1, int(r.get("Grid_Size_Z") or 1))
            except (KeyError, ValueError):
                continue
            acc[(name, grid)].append(t)
    rows = [(n, g, len(v), sum(v) / len(v), min(v), max(v), sum(v)) for (n, g), v in acc.items()]
    rows.sort(key=lambda r: -r[6])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "GridSize", "Calls", "AverageNs", "MinNs", "MaxNs", "TotalDurationNs"])
        for r in rows:
            if "rtbhip" in r[0]:
                w.writerow([r[0], r[1], r[2], "%.1f" % r[3], r[4], r[5], r[6]])
    for r in rows[:14]:
        print("%-110.110s grid %10d calls %5d avg %12.1f ns" % r[:4])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

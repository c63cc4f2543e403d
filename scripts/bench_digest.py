#!/usr/bin/env python3
"""scripts/bench_digest.py FILE -- a few lines of what a bench.py JSON line says (for the tail gpurun prints)."""
import json
import sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    rf = d["roofline"]
    print("layout %s frac %.3f  kernel_avg_ms %.4f  ms_per_step %.4f  value %.4g" % (rf.get("layout"), rf["frac"], rf["kernel_avg_ms"], d["ms_per_step"], d["value"]))
    for k, v in (rf.get("layouts") or {}).items():
        if isinstance(v, dict):
            print("  layout", k, {a: round(b, 4) for a, b in v.items()})
    for k, v in d.get("secondary", {}).items():
        if isinstance(v, dict):
            print(k, {a: (round(b, 5) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "kernel_avg_ms", "packed_layout_ms", "seconds", "error", "success_rate")},
                  "frac=%.3f" % v["roofline"]["frac"] if "roofline" in v else "", v.get("parity") if not isinstance(v.get("parity"), str) else "")
        else:
            print(k, v)
    for k in ("gather_ms", "all_gather_ms", "world"):
        if k in d:
            print(k, d[k])
except Exception as e:          # noqa: BLE001
    print("bench line unreadable:", repr(e))

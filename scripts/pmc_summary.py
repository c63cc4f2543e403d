#!/usr/bin/env python3
"""scripts/pmc_summary.py -- turn rocprofv3 --pmc counter_collection CSVs into the per-launch HBM
traffic figure bench.py reports as roofline.traffic.

    python scripts/pmc_summary.py --kernel k_kin_reg --fetch <csv> --write <csv> \
        [--calib-fetch <csv> --calib-write <csv> --calib-kernel probe_rw --calib-bytes-read B --calib-bytes-write B] \
        --out profiles/r01_pmc.json

Units and corrections (MI355X_MICROARCH.md, section HBM):
  * FETCH_SIZE / WRITE_SIZE are reported in KiB (x1024 -> bytes).
  * On gfx950 FETCH_SIZE counts 128-byte read requests at 64 bytes: it reports exactly half of a
    streaming read -> x2.  WRITE_SIZE is "uncalibrated" in the guide, so when a calibration pass
    over scripts/roofline_probe.hip (same access pattern, known byte counts) is supplied the measured
    counter/bytes ratios replace the defaults (2.0 for fetch, 1.0 for write).
FETCH_SIZE and WRITE_SIZE need separate passes (3 + 2 of the 4 TCC slots).
"""
import argparse
import collections
import csv
import json


def per_kernel(path, kernel, counter):
    vals = []
    for r in csv.DictReader(open(path)):
        if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
            vals.append(float(r["Counter_Value"]))
    return vals


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--calib-fetch")
    ap.add_argument("--calib-write")
    ap.add_argument("--calib-kernel", default="probe")
    ap.add_argument("--calib-bytes-read", type=float)
    ap.add_argument("--calib-bytes-write", type=float)
    ap.add_argument("--algorithmic-bytes", type=float, default=None)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    f = per_kernel(a.fetch, a.kernel, "FETCH_SIZE")
    w = per_kernel(a.write, a.kernel, "WRITE_SIZE")
    fetch_raw = mean(f) * 1024.0
    write_raw = mean(w) * 1024.0
    kf, kw, src = 2.0, 1.0, "guide defaults (FETCH_SIZE x2 on gfx950, WRITE_SIZE x1)"
    calib = None
    if a.calib_fetch and a.calib_write and a.calib_bytes_read and a.calib_bytes_write:
        cf = mean(per_kernel(a.calib_fetch, a.calib_kernel, "FETCH_SIZE"))
        cw = mean(per_kernel(a.calib_write, a.calib_kernel, "WRITE_SIZE"))
        if cf and cw:
            kf = a.calib_bytes_read / (cf * 1024.0)
            kw = a.calib_bytes_write / (cw * 1024.0)
            src = "calibrated on scripts/roofline_probe.hip (same access pattern, known bytes)"
            calib = {"probe_fetch_counter_bytes": cf * 1024.0, "probe_write_counter_bytes": cw * 1024.0,
                     "probe_bytes_read": a.calib_bytes_read, "probe_bytes_written": a.calib_bytes_write}
    out = collections.OrderedDict()
    out["kernel"] = a.kernel
    out["launches_sampled"] = {"fetch": len(f), "write": len(w)}
    out["FETCH_SIZE_bytes_raw_per_launch"] = fetch_raw
    out["WRITE_SIZE_bytes_raw_per_launch"] = write_raw
    out["fetch_correction"] = kf
    out["write_correction"] = kw
    out["correction_source"] = src
    if calib:
        out["calibration"] = calib
    out["hbm_read_bytes_per_launch"] = fetch_raw * kf
    out["hbm_write_bytes_per_launch"] = write_raw * kw
    out["hbm_bytes_per_launch"] = fetch_raw * kf + write_raw * kw
    if a.algorithmic_bytes:
        out["algorithmic_bytes_per_launch"] = a.algorithmic_bytes
        out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / a.algorithmic_bytes
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""scripts/placement_pmc.py [--sets S] [--launches K] [--alloc torch|hip|ext:<flags>|align:<MiB>] -- the packed fkine + jacob0 kernel (ONE
non-temporal output stream of 464 MB per launch: the single-array form of the placement lottery, profiles/r05_layout.txt: 79 or 92 us by where the
allocator put the array) on S fresh output arrays, K launches each in allocation order, all arrays alive.  Prints one JSON line per array (index,
device address, sustained us) and a summary.  Under
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -o pl -- python scripts/placement_pmc.py --launches 12
dispatch g*K .. g*K + K - 1 of the packed kernel belong to array g:  `--digest DIR RUN.jsonl`  tabulates the counters and the traced durations per
array (slow arrays beside fast ones).
--alloc: where an array comes from -- torch's caching allocator (what a torch user hands the library), a plain hipMalloc, hipExtMallocWithFlags with
the given flag word, or a hipMalloc over-allocated and aligned up to a multiple of <MiB> MiB."""
import argparse, collections, csv, ctypes as C, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--launches", type=int, default=12)
ap.add_argument("--alloc", default="torch")
ap.add_argument("--digest", nargs=2)
a = ap.parse_args()

if a.digest:
    d, runf = a.digest
    run = [json.loads(l) for l in open(runf) if l.strip().startswith("{")]
    arrays = [r for r in run if "index" in r]
    K = [r for r in run if "summary" in r][0]["launches"]
    per = collections.defaultdict(lambda: collections.defaultdict(list))     # array -> counter -> values
    order = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_kin_reg" in r["Kernel_Name"] or "k_kin_packed" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        for i, r in enumerate(rows):
            order[r["Dispatch_Id"]] = i
            per[i // K]["traced_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Dispatch_Id"] in order:
                per[order[r["Dispatch_Id"]] // K][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for g in per.values() for c in g})
    print("%-3s %-14s %-9s " % ("set", "address", "plain_us") + " ".join("%18s" % n[-18:] for n in names))
    for g in sorted(per):
        if g >= len(arrays):
            continue
        vals = []
        for n in names:
            v = per[g][n][1:] or per[g][n]        # (the first launch on an array is its first touch)
            vals.append("%18.6g" % (sum(v) / len(v)) if v else "%18s" % "-")
        print("%-3d %-14s %-9.2f " % (g, hex(arrays[g]["address"]), arrays[g]["us"]) + " ".join(vals))
    sys.exit(0)

import numpy as np
import torch
import rtbhip
from benchlib import sustained_ms
N = a.n
ets = rtbhip.models.Panda().ets()
lib, h = rtbhip.lib(), ets._handle()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
q = torch.from_numpy(np.random.default_rng(0).uniform(-np.pi, np.pi, (N, 7))).cuda()
qp = C.c_void_p(q.data_ptr())
hip = C.CDLL("libamdhip64.so")
BYTES = N * 58 * 8
keep = []


def alloc():
    if a.alloc == "torch":
        t = torch.empty((N, 58), dtype=torch.float64, device="cuda")
        keep.append(t)
        return t.data_ptr()
    p = C.c_void_p()
    if a.alloc == "hip":
        assert hip.hipMalloc(C.byref(p), C.c_size_t(BYTES)) == 0
        return p.value
    if a.alloc.startswith("ext:"):
        assert hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(BYTES), C.c_uint(int(a.alloc[4:], 0))) == 0, "hipExtMallocWithFlags refused the flags"
        return p.value
    if a.alloc.startswith("align:"):
        al = int(a.alloc[6:]) << 20
        assert hip.hipMalloc(C.byref(p), C.c_size_t(BYTES + al)) == 0
        return (p.value + al - 1) // al * al
    if a.alloc.startswith("pre:"):                      # torch arrays after a kept <GiB> GiB block: is "the first gigabyte of the process" what is fast?
        if not keep:
            assert hip.hipMalloc(C.byref(p), C.c_size_t(int(float(a.alloc[4:]) * (1 << 30)))) == 0
            keep.append(p.value)
        t = torch.empty((N, 58), dtype=torch.float64, device="cuda")
        keep.append(t)
        return t.data_ptr()
    if a.alloc.startswith("vmm:"):                      # one virtual range backed by physical chunks of <MiB> MiB created one by one (hipMemCreate / hipMemMap)
        chunk = int(a.alloc[4:]) << 20

        class Loc(C.Structure):
            _fields_ = [("type", C.c_int), ("id", C.c_int)]

        class Prop(C.Structure):
            _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", Loc), ("win32", C.c_void_p), ("compression", C.c_ubyte), ("rdma", C.c_ubyte),
                        ("usage", C.c_ushort)]

        class Acc(C.Structure):
            _fields_ = [("location", Loc), ("flags", C.c_int)]
        prop = Prop(type=1, handle_type=0, location=Loc(1, torch.cuda.current_device()))
        gran = C.c_size_t()
        assert hip.hipMemGetAllocationGranularity(C.byref(gran), C.byref(prop), C.c_int(1)) == 0      # recommended granularity
        chunk = max(chunk, gran.value) // gran.value * gran.value
        total = (BYTES + chunk - 1) // chunk * chunk
        assert hip.hipMemAddressReserve(C.byref(p), C.c_size_t(total), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)) == 0
        for off in range(0, total, chunk):
            hnd = C.c_void_p()
            assert hip.hipMemCreate(C.byref(hnd), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)) == 0
            assert hip.hipMemMap(C.c_void_p(p.value + off), C.c_size_t(chunk), C.c_size_t(0), hnd, C.c_ulonglong(0)) == 0
        acc = Acc(location=Loc(1, torch.cuda.current_device()), flags=3)
        assert hip.hipMemSetAccess(p, C.c_size_t(total), C.byref(acc), C.c_size_t(1)) == 0
        if not keep:
            keep.append(("granularity", gran.value))
            print(json.dumps({"vmm_granularity": gran.value, "chunk": chunk}), flush=True)
        return p.value
    raise SystemExit("--alloc?")


addrs = [alloc() for _ in range(a.sets)]
fns = [(lambda p=C.c_void_p(ad): lib.rtbhip_fkine_jacob_packed(h, qp, N, None, None, 0, p, 1, stream)) for ad in addrs]
# counted / traced part: K launches per array, in allocation order
for f in fns:
    for _ in range(a.launches):
        assert f() == 0
torch.cuda.synchronize()
us = []
for i, f in enumerate(fns):
    ms, _, _ = sustained_ms(f)
    us.append(ms * 1e3)
    print(json.dumps({"index": i, "address": addrs[i], "us": round(ms * 1e3, 2), "alloc": a.alloc}), flush=True)
fast = sum(1 for u in us if u <= min(us) * 1.04)
print(json.dumps({"summary": a.alloc, "launches": a.launches, "sets": a.sets, "min_us": round(min(us), 2), "max_us": round(max(us), 2), "fast_of": [fast, len(us)],
                  "frac_min": round(520.0 * N / (max(us) * 1e-6) / 8e12, 4), "frac_max": round(520.0 * N / (min(us) * 1e-6) / 8e12, 4)}), flush=True)

#!/usr/bin/env python3
"""scripts/layout_probe.py [--n N] [--fleet] -- packed ([T | J] rows, ONE output array: rtbhip_fkine_jacob_packed) against the two-array form
(rtbhip_fkine_jacob) of the headline kernel under what decided the two-array form's speed in round 4 (profiles/r04_headline_stores.txt): WHERE
the allocator put the outputs.  For each layout: eight fresh output sets (the earlier ones stay allocated, so every set sits somewhere else), the
sustained time on each; then three of them in rotation.  --fleet: the same for the 16-arm fleet call (four fresh sets).  One JSON line per
measurement; the last line is the summary (min / max / spread per layout)."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np          # noqa: E402
import torch                # noqa: E402
import rtbhip               # noqa: E402
from benchlib import sustained_ms, HBM_PEAK_GBS      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--sets", type=int, default=8)
ap.add_argument("--fleet", action="store_true")
ap.add_argument("--tag", default="")
a = ap.parse_args()
N = a.n
ets = rtbhip.models.Panda().ets()
lib, h = rtbhip.lib(), ets._handle()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
q = torch.from_numpy(np.random.default_rng(0).uniform(-np.pi, np.pi, (N, 7))).cuda()
qp = C.c_void_p(q.data_ptr())
frac = lambda ms, b=520.0 * N: b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS


def make(layout):
    if layout == "packed":
        TJ = torch.empty((N, 58), dtype=torch.float64, device="cuda")
        p = C.c_void_p(TJ.data_ptr())
        return (TJ,), (lambda: lib.rtbhip_fkine_jacob_packed(h, qp, N, None, None, 0, p, 1, stream))
    T = torch.empty((N, 4, 4), dtype=torch.float64, device="cuda")
    J = torch.empty((N, 6, 7), dtype=torch.float64, device="cuda")
    tp, jp = C.c_void_p(T.data_ptr()), C.c_void_p(J.data_ptr())
    return (T, J), (lambda: lib.rtbhip_fkine_jacob(h, qp, N, None, None, 0, tp, jp, 1, stream))


summary = {"tag": a.tag, "n": N, "lib": os.path.basename(rtbhip._lib.LIB_PATH)}
for layout in ("packed", "two", "packed", "two"):          # each layout twice, interleaved: drift shows as a difference between the passes
    sets = [make(layout) for _ in range(a.sets)]
    us = []
    for bufs, f in sets:
        assert f() == 0
        ms, _, _ = sustained_ms(f)
        us.append(round(ms * 1e3, 2))
    k = {"i": 0}

    def rot():
        sets[k["i"] % 3][1]()
        k["i"] += 1
    rot()
    rms, _, _ = sustained_ms(rot)
    rec = {"layout": layout, "tag": a.tag, "fresh_sets_us": us, "min_us": min(us), "max_us": max(us), "spread_pct": round(100 * (max(us) - min(us)) / min(us), 2),
           "rotating3_us": round(rms * 1e3, 2), "frac_min": round(frac(max(us) * 1e-3), 4), "frac_max": round(frac(min(us) * 1e-3), 4), "frac_rotating3": round(frac(rms), 4)}
    print(json.dumps(rec), flush=True)
    summary.setdefault(layout, []).append({kk: rec[kk] for kk in ("min_us", "max_us", "spread_pct", "rotating3_us", "frac_min", "frac_max", "frac_rotating3")})
    del sets
    torch.cuda.empty_cache()

if a.fleet:
    from rtbhip import urdf
    chs = [urdf.load(nm).ets() for nm in urdf.FLEET16]
    qs = []
    for i, c in enumerate(chs):
        ql = torch.from_numpy(np.clip(c.qlim, -2 * np.pi, 2 * np.pi)).cuda()
        g = torch.Generator(device="cuda").manual_seed(4 + i)
        qs.append(ql[0] + (ql[1] - ql[0]) * torch.rand((N, c.n), dtype=torch.float64, device="cuda", generator=g))
    byts = sum(N * (8 * c.n + 128 + 48 * c.n) for c in chs)
    for layout in ("packed", "two", "packed", "two"):
        fn = rtbhip.fleet_fkine_jacob_packed if layout == "packed" else rtbhip.fleet_fkine_jacob
        outs = [fn(chs, qs) for _ in range(4)]
        us = []
        for o in outs:
            f = lambda o=o: fn(chs, qs, out=o)
            f()
            ms, _, _ = sustained_ms(f)
            us.append(round(ms * 1e3, 1))
        rec = {"fleet16": layout, "tag": a.tag, "fresh_sets_us": us, "min_us": min(us), "max_us": max(us), "spread_pct": round(100 * (max(us) - min(us)) / min(us), 2),
               "frac_min": round(frac(max(us) * 1e-3, byts), 4), "frac_max": round(frac(min(us) * 1e-3, byts), 4)}
        print(json.dumps(rec), flush=True)
        summary.setdefault("fleet16_" + layout, []).append({kk: rec[kk] for kk in ("min_us", "max_us", "spread_pct", "frac_min", "frac_max")})
        del outs
        torch.cuda.empty_cache()
print(json.dumps({"summary": summary}), flush=True)

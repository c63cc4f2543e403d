#!/usr/bin/env python3
"""scripts/fleet_probe.py -- where config 5's time goes: every chain of the 17-chain fleet on its own (rtbhip_fkine_jacob, sustained) beside the one
fleet call; bytes per chain as benchsecondary.fleet_config5 prices them.  One JSON line per chain, then the sums."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from rtbhip import urdf
from benchlib import sustained_ms
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
robots = [urdf.load(nm) for nm in urdf.FLEET16]
chs = [r.ets() for r in robots]
qs = []
for i, c in enumerate(chs):
    ql = torch.from_numpy(np.clip(c.qlim, -2 * np.pi, 2 * np.pi)).cuda()
    g = torch.Generator(device="cuda").manual_seed(4 + i)
    qs.append(ql[0] + (ql[1] - ql[0]) * torch.rand((N, c.n), dtype=torch.float64, device="cuda", generator=g))
tot_ms, tot_b = 0.0, 0
for nm, c, q in zip(urdf.FLEET16, chs, qs):
    def one():
        c.fkine_jacob0(q)          # (two output arrays from the caching allocator: the same blocks every call)
    one()
    ms = sustained_ms(one)[0]
    b = N * (8 * c.n + 128 + 48 * c.n)
    tot_ms += ms; tot_b += b
    print(json.dumps({"chain": nm, "n": c.n, "ms": round(ms, 4), "GBs": round(b / ms / 1e6, 1), "frac": round(b / ms / 1e6 / 8000, 3)}), flush=True)
hold = rtbhip.fleet_fkine_jacob(chs, qs)
def fleet():
    rtbhip.fleet_fkine_jacob(chs, qs, out=hold)
fleet()
ms = sustained_ms(fleet)[0]
print(json.dumps({"sum_of_single_calls_ms": round(tot_ms, 4), "fleet16_call_ms": round(ms, 4), "bytes": tot_b, "fleet_frac": round(tot_b / ms / 1e6 / 8000, 3),
                  "singles_frac": round(tot_b / tot_ms / 1e6 / 8000, 3)}), flush=True)

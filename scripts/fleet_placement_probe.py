#!/usr/bin/env python3
"""scripts/fleet_placement_probe.py -- the 16-arm fleet call (BASELINE config 5: 16 URDF arms x 1e6 configurations, one call) on K separately allocated sets
of its 32 output arrays in one process: does its time depend on where they lie?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from rtbhip import urdf
from rtbhip.fleet import fleet_fkine_jacob
from benchlib import sustained_ms
N, K = int(os.environ.get("PROBE_N", 1000000)), int(os.environ.get("PROBE_K", 5))
arms = [urdf.load(n) for n in urdf.FLEET16]
chains = [a.ets() for a in arms]
rng = np.random.default_rng(5)
qs = [torch.from_numpy(rng.uniform(-1.5, 1.5, (N, c.n))).cuda() for c in chains]
byts = sum(8 * c.n + 128 + 48 * c.n for c in chains) * N
times = []
for k in range(K):
    out = fleet_fkine_jacob(chains, qs)
    f = lambda: fleet_fkine_jacob(chains, qs, out=out)
    f(); ms, _, _ = sustained_ms(f); times.append(round(ms, 4))
    keep = out if k == 0 else keep
    globals()["hold%d" % k] = out            # every set stays allocated
print(json.dumps({"N": N, "ms_per_output_set": times, "frac_hbm": [round(byts / (t * 1e-3) / 8e12, 3) for t in times]}))

#!/usr/bin/env python3
"""scripts/kin_scaling_probe.py -- fkine-only / jacob0-only / fkine+jacob0 at N = 2.5e5 ... 1.6e7 (sustained timing): how much of the distance between
the 1e6 lines and the streaming ceiling is launch ramp and tail."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
e = rtbhip.models.Panda().ets()
for N in (250000, 1000000, 4000000, 16000000):
    q = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7))).cuda()
    for name, fn, byts in (("fkine", lambda: e.eval(q), 184), ("jacob0", lambda: e.jacob0(q), 392), ("fkine_jacob0", lambda: e.fkine_jacob0(q), 520)):
        fn(); ms, _, _ = sustained_ms(fn)
        print(json.dumps({"what": name, "N": N, "sustained_ms": round(ms, 5), "frac_hbm": byts * N / (ms * 1e-3) / 8e12}), flush=True)
    del q

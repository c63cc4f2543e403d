#!/usr/bin/env python3
"""scripts/ik_probe_slimit.py -- a TIMING PROBE (results change with slimit; only the clock is read): how much of the config-3 launch is the
first chunk, how much the later chunks?  ik_LM over the 1e5 bench targets with slimit = 8 (the flat schedule's first chunk is the whole range),
16, 24, 40, 100 (the default), sustained timing, plus the per-call useful iterations."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ets = rtbhip.models.Panda().ets(); ets.qlim = rtbhip.models.PANDA_QLIM
Tep = ets.eval(torch.from_numpy(np.random.default_rng(1).uniform(ets.qlim[0], ets.qlim[1], (N, 7))).cuda())
for flat in (1, 0):
    rtbhip.tune("ik_flat", flat)
    for sl in (4, 8, 16, 24, 40, 64, 100):
        res = {}
        def run():
            res["o"] = ets.ik_LM(Tep, seed=2, slimit=sl)
        run()
        ms, reps, warm = sustained_ms(run)
        q, ok, it, se, E = res["o"]
        print(json.dumps({"ik_flat": flat, "slimit": sl, "sustained_ms": round(ms, 4), "success": float(ok.float().mean()), "useful_lane_iterations": int(it.sum()),
                          "ms_per_1e6_useful_lane_iterations": ms / (float(it.sum()) / 1e6)}), flush=True)
rtbhip.tune("ik_flat", 1)

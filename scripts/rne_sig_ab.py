#!/usr/bin/env python3
"""scripts/rne_sig_ab.py -- the structure-signature instantiations of k_rne / k_dyn (csrc/rne_device.h: kRneSig*) against the general kernels ON ONE
BOX: rtbhip_tune("rne_sig", 1 / 0) alternately, sustained timings (benchlib.sustained_ms) of DHRobot.rne / gravload / inertia / coriolis / accel,
plus the largest deviation between the two builds' results.  One JSON line per robot, term and batch size."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms

ap = argparse.ArgumentParser()
ap.add_argument("--robots", default="Panda,Puma560")
ap.add_argument("--n", default="1250000,10000000")
ap.add_argument("--terms", default="rne,gravload,inertia,coriolis,accel")
ap.add_argument("--rounds", type=int, default=2)
args = ap.parse_args()
for name in args.robots.split(","):
    rob = getattr(rtbhip.models.DH, name)()
    for N in [int(x) for x in args.n.split(",")]:
        rng = np.random.default_rng(7)
        q, qd, tq = (torch.from_numpy(x).cuda() for x in (rng.uniform(-3, 3, (N, rob.n)), rng.normal(size=(N, rob.n)), rng.normal(size=(N, rob.n))))
        terms = {"rne": lambda: rob.rne(q, qd, tq), "gravload": lambda: rob.gravload(q), "inertia": lambda: rob.inertia(q),
                 "coriolis": lambda: rob.coriolis(q, qd), "accel": lambda: rob.accel(q, qd, tq)}
        for term in args.terms.split(","):
            if N > 2000000 and term not in ("rne", "gravload"):
                continue
            fn = terms[term]
            ms, out = {1: [], 0: []}, {}
            for r in range(args.rounds):
                for sig in (1, 0):
                    rtbhip.tune("rne_sig", sig)
                    out[sig] = fn()
                    ms[sig].append(sustained_ms(fn)[0])
            rtbhip.tune("rne_sig", 1)
            a, b = out[1], out[0]
            dev = float((a - b).abs().max() / max(1.0, float(b.abs().max())))
            print(json.dumps({"robot": name, "term": term, "n": N, "sig_ms": round(min(ms[1]), 4), "general_ms": round(min(ms[0]), 4),
                              "speedup": round(min(ms[0]) / min(ms[1]), 3), "max_rel_deviation": dev}), flush=True)
            del out, a, b

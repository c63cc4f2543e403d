#!/usr/bin/env python3
"""scripts/tree_sig_ab.py -- the structure-signature instantiations of k_tree_rne / k_tree_dyn (csrc/tree_device.h: kTreeSig*) against the general
kernels ON ONE BOX: rtbhip_tune("tree_sig", 1 / 0) alternately, sustained timings (benchlib.sustained_ms) of rne, gravload, inertia, coriolis and
accel at N configurations, plus the largest deviation between the two builds' results.  One JSON line per robot and term."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from rtbhip import urdf
from benchlib import sustained_ms

ap = argparse.ArgumentParser()
ap.add_argument("--robots", default="UR5,UR10,wx250,Puma560")
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--rounds", type=int, default=2)
args = ap.parse_args()
for name in args.robots.split(","):
    er = urdf.load(name).erobot()
    rng = np.random.default_rng(7)
    q, qd, tq = (torch.from_numpy(x).cuda() for x in (rng.uniform(-3, 3, (args.n, er.n)), rng.normal(size=(args.n, er.n)), rng.normal(size=(args.n, er.n))))
    terms = {"rne": lambda: er.rne(q, qd, tq), "gravload": lambda: er.gravload(q), "inertia": lambda: er.inertia(q),
             "coriolis": lambda: er.coriolis(q, qd), "accel": lambda: er.accel(q, qd, tq)}
    for term, fn in terms.items():
        ms = {1: [], 0: []}
        out = {}
        for r in range(args.rounds):
            for sig in (1, 0):
                rtbhip.tune("tree_sig", sig)
                out[sig] = fn()
                ms[sig].append(sustained_ms(fn)[0])
        rtbhip.tune("tree_sig", 1)
        a, b = out[1], out[0]
        ok = torch.isfinite(b)
        dev = float(((a - b).abs()[ok]).max() / max(1.0, float(b.abs()[ok].max()))) if ok.any() else None
        print(json.dumps({"robot": name, "groups": er.n, "term": term, "n": args.n, "sig_ms": round(min(ms[1]), 4), "general_ms": round(min(ms[0]), 4),
                          "speedup": round(min(ms[0]) / min(ms[1]), 3), "max_rel_deviation": dev}), flush=True)

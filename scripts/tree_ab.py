#!/usr/bin/env python3
"""scripts/tree_ab.py -- k_tree_dyn (Dynamics.inertia / accel / coriolis of URDF arms) under one library (RTBHIP_LIB), sustained timing; one JSON
line per (robot, term) with a digest of the result and its distance from the oracle on the first rows.  Run once per library, interleaved."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from rtbhip import urdf
from benchlib import sustained_ms
N = int(os.environ.get("TREE_AB_N", 1000000))
tag = os.environ.get("TREE_AB_TAG", os.path.basename(os.environ.get("RTBHIP_LIB", "shipped")))
for name in ("UR5", "Panda"):
    er = urdf.load(name).erobot()
    rng = np.random.default_rng(7)
    q, qd, tq = (torch.from_numpy(x).cuda() for x in (rng.uniform(-3, 3, (N, er.n)), rng.normal(size=(N, er.n)), rng.normal(size=(N, er.n))))
    for term, fn in (("inertia", lambda: er.inertia(q)), ("accel", lambda: er.accel(q, qd, tq)), ("coriolis", lambda: er.coriolis(q, qd)), ("rne", lambda: er.rne(q, qd, tq))):
        out = fn(); ms, _, _ = sustained_ms(fn)
        h = out[:4096].cpu().numpy()
        print(json.dumps({"lib": tag, "robot": name, "n": er.n, "term": term, "N": N, "sustained_ms": round(ms, 5),
                          "digest": hashlib.sha1(np.ascontiguousarray(h + 0.0).tobytes()).hexdigest()[:12], "sum": float(h.sum())}), flush=True)

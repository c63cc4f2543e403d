#!/usr/bin/env python3
"""scripts/dyn_ab.py -- k_dyn (Dynamics.inertia / coriolis / accel of DH arms) under one library (RTBHIP_LIB), sustained timing; one JSON line per
(robot, term) with a digest of the first rows and, for coriolis, the distance from the oracle on 64 rows.  Run once per library, interleaved."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
from oracle import oracle
N = int(os.environ.get("DYN_AB_N", 1000000))
tag = os.environ.get("DYN_AB_TAG", "product")
for name, rob, mdh in (("Panda", rtbhip.models.DH.Panda(), 1), ("Puma560", rtbhip.models.DH.Puma560(), 0)):
    n = rob.n
    rng = np.random.default_rng(6)
    ql = np.asarray(rob.qlim)
    qh, qdh, tqh = rng.uniform(ql[0], ql[1], (N, n)), rng.normal(size=(N, n)), rng.normal(size=(N, n)) * 5
    q, qd, tq = (torch.from_numpy(x).cuda() for x in (qh, qdh, tqh))
    for term, fn in (("inertia", lambda: rob.inertia(q)), ("coriolis", lambda: rob.coriolis(q, qd)), ("accel", lambda: rob.accel(q, qd, tq))):
        out = fn(); ms, _, _ = sustained_ms(fn)
        h = out[:4096].cpu().numpy()
        line = {"lib": tag, "robot": name, "n": n, "term": term, "N": N, "sustained_ms": round(ms, 5),
                "digest": hashlib.sha1(np.ascontiguousarray(h + 0.0).tobytes()).hexdigest()[:12]}
        if term == "coriolis":
            ref = oracle.coriolis_dh(rob.L24(), mdh, qh[:64], qdh[:64])
            line["max_err_vs_oracle_rel"] = float(np.abs(h[:64] - ref).max() / np.abs(ref).max())
        print(json.dumps(line), flush=True)

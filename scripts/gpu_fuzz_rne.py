#!/usr/bin/env python3
"""scripts/gpu_fuzz_rne.py -- every size of the DH inverse-dynamics kernels on the device against the C restatement of core/ne.c (oracle.rne_dh): random
standard- and modified-DH chains of 1..24 joints, all-revolute and with prismatic joints, zero / nonzero centres of mass, diagonal / full inertia
tensors, friction and motor inertia; the call forms rne(q, qd, qdd), qd = None (gravload / itorque), a gravity vector, a tip wrench.  130 rows (three
tiles).  One JSON line per joint count; exit code 1 on a miss (> 1e-10 of the torques' scale)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import rtbhip
from oracle import oracle

miss, t0 = [], time.time()
for n in range(1, 25):
    worst = 0.0
    for rep in range(4):
        rng = np.random.default_rng(11000 + 10 * n + rep)
        mdh, allrev = rep % 2, rep < 2
        links = []
        for j in range(n):
            I = rng.uniform(0.01, 0.1, 3)
            Ifull = np.diag(I) if rng.uniform() < 0.4 else np.diag(I) + 0.002 * (np.ones((3, 3)) - np.eye(3))
            kw = dict(a=float(rng.choice([0.0, 0.05 + 0.02 * j])), alpha=float(rng.choice([0.0, np.pi / 2, -np.pi / 2, 0.3])), m=1.0 + 0.1 * j,
                      r=[0.0, 0.0, 0.0] if rng.uniform() < 0.3 else list(rng.uniform(-0.05, 0.05, 3)), I=Ifull, Jm=1e-4 * j, G=1.0 + j, B=1e-3, Tc=[0.01, -0.02])
            pris = (not allrev) and rng.uniform() < 0.3
            if mdh:
                links.append(rtbhip.PrismaticMDH(theta=0.3, qlim=[0.0, 0.4], **kw) if pris else rtbhip.RevoluteMDH(d=float(rng.choice([0.0, 0.1])), **kw))
            else:
                links.append(rtbhip.PrismaticDH(theta=0.3, qlim=[0.0, 0.4], **kw) if pris else rtbhip.RevoluteDH(d=float(rng.choice([0.0, 0.1])), **kw))
        rob = rtbhip.DHRobot(links)
        L = rob.L24()
        q, qd, qdd = rng.uniform(-1, 1, (130, n)), rng.normal(size=(130, n)), rng.normal(size=(130, n))
        z = np.zeros_like(q)
        fext = rng.normal(size=6)
        for name, args, kw, oargs in (("rne", (q, qd, qdd), {}, (q, qd, qdd, rob._gravity_c(None), None)),
                                      ("rne gravity", (q, qd, qdd), {"gravity": [1.0, -2.0, 9.0]}, (q, qd, qdd, rob._gravity_c([1.0, -2.0, 9.0]), None)),
                                      ("rne fext", (q, qd, qdd), {"fext": fext}, (q, qd, qdd, rob._gravity_c(None), fext)),
                                      ("gravload", (q, None, None), {}, (q, z, z, rob._gravity_c(None), None)),
                                      ("itorque", (q, None, qdd), {"gravity": [0, 0, 0]}, (q, z, qdd, rob._gravity_c([0, 0, 0]), None))):
            got = rob.rne(*args, **kw)
            want = oracle.rne_dh(L, mdh, *oargs)
            d = float(np.abs(got - want).max() / max(1.0, np.abs(want).max()))
            worst = max(worst, d)
            if not d <= 1e-10:
                miss.append([n, mdh, allrev, name, d])
    print(json.dumps({"joints": n, "worst_relative_deviation": worst}), flush=True)
print(json.dumps({"misses": miss[:40], "n_misses": len(miss), "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if miss else 0)

#!/usr/bin/env python3
"""scripts/gpu_fuzz_dyn.py -- the dynamics kernels of every size on the device against the oracle, on random robots: link trees of 1..20 joints
(numbered automatically and by hand) through k_tree_rne / k_tree_dyn, DH and modified-DH chains of 1..16 joints with prismatic joints through k_rne /
k_dyn.  Two rows each (the oracle is Python).  One JSON line per family with the worst relative deviation; exit code 1 on a miss."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import rtbhip
from rtbhip import ERobot
from oracle import oracle, erobot as oer
from test_erobot_rne import random_tree, dfs
from test_erobot_dynamics import renumbered_case

def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))

worst, miss, t0 = {}, [], time.time()
def note(fam, key, v, tol):
    worst[fam] = max(worst.get(fam, 0.0), v)
    if not v <= tol:
        miss.append((fam, key, v))

g = np.array([0.5, -0.3, -9.81])
count = {"tree": 0, "tree_hand_numbered": 0, "dh": 0}
seen = set()
for seed in range(400):
    rng = np.random.default_rng(1000 + seed)
    prod, orc = random_tree(rng, n_links=int(rng.integers(1, 27)))
    rob = ERobot(prod)
    n = rob.n
    if not 1 <= n <= 20 or (n, "t") in seen and count["tree"] >= 44:
        continue
    seen.add((n, "t")); count["tree"] += 1
    links = dfs(orc)
    q, qd, tq = rng.uniform(-2, 2, (66, n)), rng.normal(size=(66, n)), rng.normal(size=(66, n))
    rob.gravity = g
    k = slice(64, 66)                      # rows of the second tile
    note("tree rne", (seed, n), rel(rob.rne(q, qd, tq)[k], oer.erobot_rne(links, q[k], qd[k], tq[k], g)), 1e-10)
    note("tree gravload", (seed, n), rel(rob.gravload(q)[k], oer.erobot_rne(links, q[k], 0 * q[k], 0 * q[k], g)), 1e-10)
    note("tree itorque", (seed, n), rel(rob.itorque(q, tq)[k], oer.erobot_rne(links, q[k], 0 * q[k], tq[k], (0, 0, 0))), 1e-10)
    M = oer.erobot_inertia(links, q[k])
    note("tree inertia", (seed, n), rel(rob.inertia(q)[k], M), 1e-12)
    note("tree coriolis", (seed, n), rel(rob.coriolis(q, qd)[k], oer.erobot_coriolis(links, q[k], qd[k])), 1e-12)
    if np.linalg.cond(M).max() < 1e6:
        note("tree accel", (seed, n), rel(rob.accel(q, qd, tq)[k], oer.erobot_accel(links, q[k], qd[k], tq[k], g)), 1e-8)
for seed in range(60):
    try:
        rob, links, rng = renumbered_case(2000 + seed, 3 + seed % 12)           # (renumbered_case draws 3..9 joints)
    except Exception:
        continue
    n = rob.n
    count["tree_hand_numbered"] += 1
    q, qd, tq = rng.uniform(-2, 2, (3, n)), rng.normal(size=(3, n)), rng.normal(size=(3, n))
    rob.gravity = g
    M = oer.erobot_inertia(links, q)
    note("hand-numbered inertia", (seed, n), rel(rob.inertia(q), M), 1e-12)
    note("hand-numbered coriolis", (seed, n), rel(rob.coriolis(q, qd), oer.erobot_coriolis(links, q, qd)), 1e-12)
    note("hand-numbered rne", (seed, n), rel(rob.rne(q, qd, tq), oer.erobot_rne(links, q, qd, tq, g)), 1e-10)
    if np.linalg.cond(M).max() < 1e6:
        note("hand-numbered accel", (seed, n), rel(rob.accel(q, qd, tq), oer.erobot_accel(links, q, qd, tq, g)), 1e-8)
for seed in range(64):
    rng = np.random.default_rng(3000 + seed)
    n, mdh = 1 + seed % 16, (seed // 16) % 2
    allrev = seed >= 32
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
    count["dh"] += 1
    q, qd, tq = rng.uniform(-1, 1, (66, n)), rng.normal(size=(66, n)), rng.normal(size=(66, n))
    k = slice(64, 66)
    gc = -np.array([0.0, 0.0, -9.81])
    ref = oracle.coriolis_dh(L, mdh, q[k], qd[k])
    note("dh coriolis", (seed, n, mdh), float(np.abs(rob.coriolis(q, qd)[k] - ref).max() / max(np.abs(ref).max(), 1e-3)), 1e-10)
    Mo = oracle.inertia_dh(L, mdh, q[k])
    note("dh inertia", (seed, n, mdh), rel(rob.inertia(q)[k], Mo), 1e-10)
    if np.linalg.cond(Mo).max() < 1e6:
        ra = oracle.accel_dh(L, mdh, q[k], qd[k], tq[k], gc)
        note("dh accel", (seed, n, mdh), rel(rob.accel(q, qd, tq)[k], ra), 1e-7)
for fam in sorted(worst):
    print(json.dumps({"family": fam, "worst_relative_deviation": worst[fam]}))
print(json.dumps({"robots": count, "misses": [list(map(str, m)) for m in miss], "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if miss else 0)

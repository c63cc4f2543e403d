#!/usr/bin/env python3
"""scripts/jit_diag.py <case> -- one diagnostic case of the run-time instantiations per process (a device fault must not take the others down):
   rne [hexmask]   perturbed DH Panda: run-time k_rne (signature AND mask) against the general kernel, number of differing entries
   ik <robot>      URDF robot: run-time k_ik against the general kernel
   tree <robot>    URDF robot: run-time k_tree_rne / k_tree_dyn against the general kernels"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np
import rtbhip
from rtbhip import jit, urdf

case = sys.argv[1]
rtbhip.tune("jit", 2)
rng = np.random.default_rng(5)


def cmp(tag, a, b):
    a, b = np.asarray(a), np.asarray(b)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    return {"what": tag, "differing": int((a != b).sum()), "of": int(a.size), "max_abs": float(np.nanmax(d)) if a.size else 0.0, "nan": int(np.isnan(d).sum())}


out = {"case": sys.argv[1:]}
if case == "rne":
    p = rtbhip.models.DH.Panda()
    links = list(p.links)
    k = links[3]
    links[3] = rtbhip.RevoluteMDH(a=k.a, d=k.d, alpha=k.alpha + 0.01, m=k.m, r=k.r, I=k.I, G=1)
    rob = rtbhip.DHRobot(links)
    N = 4000
    q, qd, qdd = rng.uniform(-3, 3, (N, 7)), rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
    calls = {"rne": lambda: rob.rne(q, qd, qdd), "gravload": lambda: rob.gravload(q), "inertia": lambda: rob.inertia(q), "coriolis": lambda: rob.coriolis(q, qd),
             "accel": lambda: rob.accel(q, qd, qdd)}
    fast = {k: np.asarray(f()) for k, f in calls.items()}
    st = jit.stats()
    rtbhip.tune("rne_sig", 0)
    gen = {k: np.asarray(f()) for k, f in calls.items()}
    out["cmp"] = [cmp(k, fast[k], gen[k]) for k in calls]
    out["names"] = jit.names(rob)[0][:1]
elif case == "ik":
    e = urdf.load(sys.argv[2]).ets()
    lim = np.clip(e.qlim, -2.8, 2.8)
    T = np.asarray(e.eval(rng.uniform(lim[0], lim[1], (2000, e.n))))
    fast = e.ik_LM(T, seed=4)
    st = jit.stats()
    rtbhip.tune("ik_sig", 0)
    gen = e.ik_LM(T, seed=4)
    out["cmp"] = [cmp(n, x, y) for n, x, y in zip(("q", "success", "iterations", "searches", "residual"), fast, gen)]
    out["success_rate"] = float(np.asarray(fast[1]).mean())
    out["names"] = jit.names(e)[0][:1]
elif case == "tree":
    t = urdf.load(sys.argv[2]).erobot()
    N, n = 700, t.n
    q, qd, qdd = rng.uniform(-2, 2, (N, n)), rng.normal(size=(N, n)), rng.normal(size=(N, n))
    calls = {"rne": lambda: t.rne(q, qd, qdd), "gravload": lambda: t.gravload(q)}
    if n <= 20:
        calls.update(inertia=lambda: t.inertia(q), coriolis=lambda: t.coriolis(q, qd), accel=lambda: t.accel(q, qd, qdd))
    only = sys.argv[3:] or list(calls)
    fast = {}
    for k in only:
        print("launch", k, flush=True)
        fast[k] = np.asarray(calls[k]())
    st = jit.stats()
    rtbhip.tune("tree_sig", 0)
    gen = {k: np.asarray(calls[k]()) for k in only}
    out["cmp"] = [cmp(k, fast[k], gen[k]) for k in only]
    out["n"] = n
elif case == "chain":
    # a synthetic robot of n revolute joints in series (optionally a second branch off link `b`)
    from rtbhip import ET, ETS, Link, ERobot
    n, b = int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else -1
    links = []
    for i in range(n):
        T = np.eye(4); T[:3, 3] = rng.uniform(-0.3, 0.3, 3)
        th = rng.uniform(-1, 1); c, s_ = np.cos(th), np.sin(th)
        T[:3, :3] = np.array([[1, 0, 0], [0, c, -s_], [0, s_, c]])
        par = None if i == 0 else (links[b] if (b >= 0 and i == n - 2) else links[i - 1])
        links.append(Link(ets=ETS([ET.SE3(T), ET.Rz()]), m=float(rng.uniform(0.5, 3)), r=rng.uniform(-0.2, 0.2, 3), parent=par, name="k%d" % i))
    t = ERobot(links)
    N = 500
    q, qd, qdd = rng.uniform(-2, 2, (N, t.n)), rng.normal(size=(N, t.n)), rng.normal(size=(N, t.n))
    print("groups", t.n, flush=True)
    fast = np.asarray(t.rne(q, qd, qdd))
    st = jit.stats()
    rtbhip.tune("tree_sig", 0)
    gen = np.asarray(t.rne(q, qd, qdd))
    out["cmp"] = [cmp("rne", fast, gen)]
out["jit"] = {k: st[k] for k in ("compiled", "disk_hits", "failed", "launches", "compile_seconds", "last_error")}
print(json.dumps(out), flush=True)

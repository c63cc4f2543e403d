#!/usr/bin/env python3
"""scripts/gpu_fuzz_jit.py -- random STRUCTURED robots through their run-time instantiations (csrc/jit.cpp) against the general kernels, bit for bit.

The straight-line structure kernels claim the general kernels' bits BY CONSTRUCTION (csrc/exactform.h: every structured product is the general
product's operation sequence with exact zeros and ones rewritten away; kin_device.h mix_pp: every sum of two products downstream is written out).  The
hand-written tests check that on the robots of the model library; this script checks it on robots nobody looked at: constants drawn from the classes the
chain compiler distinguishes (identity, quarter turns, one-axis rotations, pure translations on some axes, general), all of them mixed.
    DH link tables    2..8 revolute joints, alpha in {0, +-pi/2, general}, a / d / centre of mass / friction / motor inertia zero or not     rne, gravload, inertia, coriolis, accel
    ETS chains        3..8 revolute joints about x / y / z, constants = products of structured elementary transforms                          ik_LM (q, success, iterations, searches, residual);
                      jacob0_dot, manipulability (3 methods), jacobm, analytical Jacobians (4 representations)
    link trees        3..12 groups, branching, prismatic joints, URDF-style origins (rpy multiples of pi/2 or general, xyz with zeros)      rne, inertia, coriolis, accel
Every output of the run-time kernel (rtbhip_tune "jit" = 2: a launch waits for its instantiation) must EQUAL the general kernel's (*_sig = 0).
One JSON line per family; exit code 1 on a differing bit, 0 otherwise (also 0, with a note, where libhiprtc.so is absent)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import rtbhip
from rtbhip import ET, ETS, Link, ERobot, jit

if not jit.stats()["available"]:
    print(json.dumps({"skipped": "libhiprtc.so not found: every robot takes the general kernels"}))
    sys.exit(0)
rtbhip.tune("jit", 2)
rtbhip.tune("jit_eager", 0)            # only what is launched gets compiled
miss, t0 = [], time.time()
H = np.pi / 2


def both(key, call):
    """call() on the robot's own instantiation, then on the general kernel; returns the number of arrays compared"""
    l0 = jit.stats()["launches"]
    fast = [np.asarray(x).copy() for x in call()]
    served = jit.stats()["launches"] - l0
    rtbhip.tune(key, 0)
    try:
        gen = [np.asarray(x) for x in call()]
    finally:
        rtbhip.tune(key, 1)
    return fast, gen, served


# ---------------------------------------------------------------- DH link tables
line = {"family": "DH link tables (k_rne, k_rne_atrest, k_dyn)", "robots": 0, "arrays": 0, "served_by_jit": 0}
for n in range(2, 9):
    rng = np.random.default_rng(21000 + n)
    mdh = n % 2
    links = []
    for j in range(n):
        I = rng.uniform(0.01, 0.1, 3)
        kw = dict(a=float(rng.choice([0.0, 0.05 + 0.02 * j])), alpha=float(rng.choice([0.0, H, -H, 0.3])), d=float(rng.choice([0.0, 0.1])), m=1.0 + 0.1 * j,
                  r=[0.0, 0.0, 0.0] if rng.uniform() < 0.4 else list(rng.uniform(-0.05, 0.05, 3)),
                  I=np.diag(I) if rng.uniform() < 0.5 else np.diag(I) + 0.002 * (np.ones((3, 3)) - np.eye(3)))
        if n % 3 == 0:
            kw.update(Jm=1e-4 * (j + 1), G=1.0 + j, B=1e-3, Tc=[0.01, -0.02])
        links.append(rtbhip.RevoluteMDH(**kw) if mdh else rtbhip.RevoluteDH(**kw))
    rob = rtbhip.DHRobot(links)
    if not jit.names(rob)[0]:
        continue                       # (a table that happens to match a built-in instantiation)
    q, qd, qdd = rng.uniform(-2, 2, (130, n)), rng.normal(size=(130, n)), rng.normal(size=(130, n))
    fast, gen, served = both("rne_sig", lambda: (rob.rne(q, qd, qdd), rob.gravload(q), rob.inertia(q), rob.coriolis(q, qd), rob.accel(q, qd, qdd)))
    line["robots"] += 1; line["arrays"] += len(fast); line["served_by_jit"] += served
    for name, a, b in zip(("rne", "gravload", "inertia", "coriolis", "accel"), fast, gen):
        if not np.array_equal(a, b, equal_nan=True):
            miss.append(["dh", n, name, int((a != b).sum()), float(np.nanmax(np.abs(a - b)))])
print(json.dumps(line), flush=True)

# ---------------------------------------------------------------- ETS chains (IK)
line = {"family": "ETS chains (k_ik)", "robots": 0, "arrays": 0, "served_by_jit": 0, "success_rate": []}
diff_line = {"family": "ETS chains (k_kin_diff)", "robots": 0, "arrays": 0, "served_by_jit": 0}
for n in range(3, 9):
    rng = np.random.default_rng(22000 + n)
    ets = ETS()
    for j in range(n):
        # a structured constant in front of every joint: translations on a random subset of axes, a quarter turn / a general turn / nothing
        for ax in ("tx", "ty", "tz"):
            if rng.uniform() < 0.45:
                ets = ets * getattr(ET, ax)(float(rng.uniform(0.05, 0.3)))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            ets = ets * getattr(ET, ["Rx", "Ry", "Rz"][int(rng.integers(0, 3))])(float(rng.choice([H, -H])))
        elif kind == 1:
            ets = ets * getattr(ET, ["Rx", "Ry", "Rz"][int(rng.integers(0, 3))])(float(rng.uniform(-1.2, 1.2)))
        elif kind == 2:
            ets = ets * ET.Rx(float(rng.uniform(-1, 1))) * ET.Rz(float(rng.uniform(-1, 1)))            # general
        ets = ets * getattr(ET, ["Rx", "Ry", "Rz"][int(rng.integers(0, 3))])()
    ets = ets * ET.tz(0.1)
    ets = ETS(list(ets))
    ets.qlim = np.array([[-2.6] * n, [2.6] * n])
    if not jit.names(ets)[0]:
        continue
    qs = rng.uniform(-2.3, 2.3, (400, n))
    Tep = np.asarray(ets.eval(qs))
    fast, gen, served = both("ik_sig", lambda: ets.ik_LM(Tep, seed=3, slimit=20))
    line["robots"] += 1; line["arrays"] += len(fast); line["served_by_jit"] += served
    line["success_rate"].append(round(float(np.asarray(fast[1]).mean()), 3))
    for name, a, b in zip(("q", "success", "iterations", "searches", "residual"), fast, gen):
        if not np.array_equal(a, b, equal_nan=True):
            miss.append(["ik", n, name, int((a != b).sum()), float(np.nanmax(np.abs(a.astype(float) - b.astype(float))))])
    # the same chain through k_kin_diff<NJ, MODE, SIG> (csrc/diff_kernel.h): jacob0_dot, manipulability in its three methods, jacobm, the analytical Jacobians
    qdv = rng.normal(size=qs.shape)
    names = ("jacob0_dot", "manip_yoshikawa", "manip_minsingular", "manip_invcondition_trans", "jacobm", "jacobm_rot", "ja_rpy/xyz", "ja_rpy/zyx", "ja_eul", "ja_exp")
    fast, gen, served = both("diff_sig", lambda: (ets.jacob0_dot(qs, qdv), ets.manipulability(qs), ets.manipulability(qs, method="minsingular"),
                                                  ets.manipulability(qs, method="invcondition", axes="trans"), ets.jacobm(qs), ets.jacobm(qs, axes="rot"),
                                                  ets.jacob0_analytical(qs, "rpy/xyz"), ets.jacob0_analytical(qs, "rpy/zyx"), ets.jacob0_analytical(qs, "eul"),
                                                  ets.jacob0_analytical(qs, "exp")))
    diff_line["robots"] += 1; diff_line["arrays"] += len(fast); diff_line["served_by_jit"] += served
    for name, a, b in zip(names, fast, gen):
        if not np.array_equal(a, b, equal_nan=True):
            miss.append(["diff", n, name, int((a != b).sum()), float(np.nanmax(np.abs(a - b)))])
print(json.dumps(line), flush=True)
print(json.dumps(diff_line), flush=True)

# ---------------------------------------------------------------- link trees
line = {"family": "link trees (k_tree_rne, k_tree_dyn)", "robots": 0, "arrays": 0, "served_by_jit": 0}


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]]); Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]]); Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


for n in (3, 4, 6, 7, 9, 12):
    rng = np.random.default_rng(23000 + n)
    links = []
    for i in range(n):
        parent = None if i == 0 else int(rng.integers(max(0, i - 3), i))
        T = np.eye(4)
        style = int(rng.integers(0, 3))
        if style == 0:
            T[:3, :3] = rpy(*(H * rng.integers(-1, 3, 3)))                        # URDF-style: multiples of pi / 2 (their cos(pi/2) stays 6.1e-17)
        elif style == 1:
            T[:3, :3] = rpy(*rng.uniform(-1, 1, 3))
        T[:3, 3] = rng.uniform(-0.3, 0.3, 3) * (rng.uniform(size=3) < 0.6)
        ax = ["Rx", "Ry", "Rz", "tz"][int(rng.integers(0, 4))]
        m = float(rng.uniform(0.2, 2))
        kw = dict(ets=ETS([ET.SE3(T), getattr(ET, ax)()]), m=m, r=rng.uniform(-0.2, 0.2, 3) * (rng.uniform() < 0.7), parent=(links[parent] if parent is not None else None),
                  name="k%d" % i)
        if rng.uniform() < 0.6:
            I = rng.uniform(0.01, 0.1, 3)
            kw["I"] = np.diag(I) + 0.002 * (np.ones((3, 3)) - np.eye(3))
        links.append(Link(**kw))
    rob = ERobot(links)
    if rob.n != n or not jit.names(rob)[0]:
        continue
    q, qd, qdd = rng.uniform(-2, 2, (130, n)), rng.normal(size=(130, n)), rng.normal(size=(130, n))
    fast, gen, served = both("tree_sig", lambda: (rob.rne(q, qd, qdd), rob.gravload(q), rob.inertia(q), rob.coriolis(q, qd), rob.accel(q, qd, qdd)))
    line["robots"] += 1; line["arrays"] += len(fast); line["served_by_jit"] += served
    for name, a, b in zip(("rne", "gravload", "inertia", "coriolis", "accel"), fast, gen):
        if not np.array_equal(a, b, equal_nan=True):
            miss.append(["tree", n, name, int((a != b).sum()), float(np.nanmax(np.abs(a - b)))])
print(json.dumps(line), flush=True)
st = jit.stats()
print(json.dumps({"misses": miss[:40], "n_misses": len(miss), "seconds": round(time.time() - t0, 1),
                  "jit": {k: st[k] for k in ("compiled", "disk_hits", "failed", "launches", "compile_seconds", "last_error")}}))
sys.exit(1 if miss or st["failed"] else 0)

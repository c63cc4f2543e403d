#!/usr/bin/env python3
"""scripts/gpu_fuzz_kin.py -- every size of the kinematics kernels and their differential consumers on the device against the oracle, on random
chains of 1..16 joints (every transform kind, flips, SE3 constants; with and without a tool): fkine, jacob0 / jacobe, hessian0 / hessiane, jacob0_dot,
manipulability (three measures, axis subsets), jacobm, the analytical Jacobians, partial_fkine0, link frames.  66 rows each (two tiles).  One JSON
line per joint count with the worst absolute deviation; exit code 1 on a miss (> 1e-9 of the result's scale)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import rtbhip
from oracle import oracle, chains
from helpers import product_ets
from test_random_chains import random_spec

miss, t0 = [], time.time()
by_name = {}
def dev(name, n, got, want, tol=1e-9):
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape:
        miss.append([n, name, "shape %s != %s" % (got.shape, want.shape)]); return 0.0
    bad = ~np.isfinite(got) & np.isfinite(want)
    d = float(np.nanmax(np.abs(got - want)) / max(1.0, float(np.nanmax(np.abs(want))))) if got.size else 0.0
    if bad.any() or not d <= tol:
        miss.append([n, name, d])
    by_name[name.split()[0]] = max(by_name.get(name.split()[0], 0.0), d)
    return d

for n in range(1, 17):
    worst = 0.0
    for rep in range(3):
        rng = np.random.default_rng(9000 + 10 * n + rep)
        spec = random_spec(rng, n)
        ets, ch = product_ets(spec), chains.Chain(spec)
        tool = None if rep == 0 else chains.elementary("tx", rng.uniform(-0.2, 0.2)) @ chains.elementary("Rx", rng.uniform(-1, 1))
        q = rng.uniform(-2.5, 2.5, (66, n)); qd = rng.normal(size=(66, n))
        k = slice(60, 66)
        worst = max(worst, dev("fkine", n, ets.eval(q, tool=tool), oracle.fkine(ch, q, tool=tool)))
        worst = max(worst, dev("jacob0", n, ets.jacob0(q, tool=tool), oracle.jacob(ch, q, tool, 0)))
        worst = max(worst, dev("jacobe", n, ets.jacobe(q, tool=tool), oracle.jacob(ch, q, tool, 1)))
        worst = max(worst, dev("hessian0", n, ets.hessian0(q, tool=tool)[k], oracle.hessian(ch, q[k], tool, 0)))
        worst = max(worst, dev("hessiane", n, ets.hessiane(q, tool=tool)[k], oracle.hessian(ch, q[k], tool, 1)))
        worst = max(worst, dev("jacob0_dot", n, ets.jacob0_dot(q, qd, tool=tool)[k], oracle.jacob_dot(ch, q[k], qd[k], tool, 0)))
        marks = sorted(set(rng.integers(0, ch.m + 1, 5).tolist()))
        worst = max(worst, dev("link_frames", n, ets.link_frames(q, marks)[k], oracle.link_frames(ch, q[k], marks)))
        for method in ("yoshikawa", "minsingular", "invcondition"):
            for axes in ("all", "trans", "rot"):
                try:
                    want = oracle.manipulability(ch, q[k], axes=axes, tool=tool, method=method)
                except Exception:
                    continue
                worst = max(worst, dev("manipulability %s %s" % (method, axes), n, ets.manipulability(q, method=method, axes=axes, tool=tool)[k], want, tol=1e-7))
        try:
            want = oracle.jacobm(ch, q[k], tool=tool)
            # (fewer than six joints: J J^T is singular, the measure is 0 and its gradient -- m times the inverse of a singular matrix -- is whatever
            #  rounding makes of it in numpy and in the LDL^T alike: not compared, profiles/r06_aj_fuzz_more.txt)
            if n >= 6 and np.isfinite(want).all() and np.abs(want).max() < 1e6:
                worst = max(worst, dev("jacobm", n, ets.jacobm(q, tool=tool)[k], want, tol=1e-6))
        except Exception:
            pass
        for r in ("rpy/xyz", "rpy/zyx", "eul", "exp"):
            try:
                want = oracle.jacob0_analytical(ch, q[k], r, tool=tool)
            except np.linalg.LinAlgError:              # a representation's singularity (a chain without rotation under "exp"): the reference raises
                continue
            if np.isfinite(want).all() and np.abs(want).max() < 1e6:
                worst = max(worst, dev("jacob0_analytical " + r, n, ets.jacob0_analytical(q, representation=r, tool=tool)[k], want, tol=1e-6))
        if n <= 10:
            for order in (2, 3):
                want = np.array([oracle.partial_fkine0(ch, row, order, tool=tool) for row in q[k]])
                worst = max(worst, dev("partial_fkine0 %d" % order, n, ets.partial_fkine0(q[k], n=order, tool=tool), want))
    print(json.dumps({"joints": n, "worst_relative_deviation": worst}), flush=True)
print(json.dumps({"worst_by_quantity": by_name}))
print(json.dumps({"misses": miss[:40], "n_misses": len(miss), "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if miss else 0)

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/dbg2.py <<'PY'
import sys, os, warnings
warnings.simplefilter("ignore")
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "robotics-toolbox-python_amd")]
import numpy as np, rtbhip
from oracle import oracle
from helpers import chain_from_ets
from rtbhip import urdf as U
rng = np.random.default_rng(14)
qs = rng.uniform(-2.5, 2.5, (40, 14))
for name in ("KinovaGen3", "Fetch"):
    e = U.load(name).ets()
    e.qlim = np.clip(e.qlim, -np.pi, np.pi)
    c2 = chain_from_ets(e)
    span = c2.qlim[1] - c2.qlim[0]
    q2 = rng.uniform(c2.qlim[0] + 0.1 * span, c2.qlim[1] - 0.1 * span, (30, e.n))
    T2 = oracle.fkine(c2, q2)
    q0 = np.clip(q2 + 0.02 * span * rng.normal(size=q2.shape), c2.qlim[0], c2.qlim[1])
    for kw in ({}, {"kq": 0.1}, {"kq": 0.1, "km": 0.1}):
        sol = e.ikine_LM(T2, q0=q0, seed=1, slimit=5, **kw)
        agree = hits = 0
        for i in range(30):
            o = oracle.ikine_py(c2, T2[i], np.array([q0[i]] + [e.ik_restart(1, i, d) for d in range(1, 5)]), step="lm", slimit=5, **kw)
            g = (int(sol.each["success"][i]), int(sol.each["iterations"][i]), int(sol.each["searches"][i]))
            same = (o[1], o[2], o[3]) == g
            agree += same
            if o[1] and o[3] == 1: hits += 1
            if i < 5: print(name, kw, i, "oracle", o[1], o[2], o[3], "gpu", g, "dq %.2e" % np.nanmax(np.abs(np.nan_to_num(sol.q[i]) - np.nan_to_num(o[0]))))
        print(name, kw, "agree", agree, "hits", hits, flush=True)
PY
timeout 300 python /tmp/dbg2.py 2>&1 | grep -v amdgpu.ids

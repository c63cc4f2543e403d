#!/bin/bash
cd $GRAFT_REPO_ROOT
PYTHONPATH=robotics-toolbox-python_amd python - <<'PY'
import numpy as np, rtbhip
from rtbhip import urdf
np.set_printoptions(precision=5, suppress=True, linewidth=200)
ur5 = urdf.load("UR5").ets()
print(ur5, ur5.n, ur5.jindices)
qt = np.array([0, -0.3, 0, -2.2, 0, 2.0])
Tep = ur5.eval(qt)
for name in ("ik_NR", "ik_GN", "ik_LM"):
    q, ok, it, se, E = getattr(ur5, name)(Tep, seed=0)
    e = rtbhip.angle_axis(ur5.eval(q), Tep)
    print(name, q, ok, it, se, E, "E(FK(q)) = %.3g" % (0.5 * e @ e))
for name in ("ikine_NR", "ikine_GN", "ikine_LM"):
    sol = getattr(ur5, name)(Tep, seed=0, **({} if name == "ikine_LM" else {"pinv": True}))
    e = rtbhip.angle_axis(ur5.eval(sol.q), Tep)
    print(name, sol.q, sol.success, sol.iterations, sol.searches, sol.residual, "E(FK(q)) = %.3g" % (0.5 * e @ e))
    sol = getattr(ur5, name)(Tep, q0=qt + 0.05, seed=0, **({} if name == "ikine_LM" else {"pinv": True}))
    e = rtbhip.angle_axis(ur5.eval(sol.q), Tep)
    print(name, "q0 near", sol.q, sol.success, sol.iterations, sol.searches, sol.residual, "E(FK(q)) = %.3g" % (0.5 * e @ e))
PY

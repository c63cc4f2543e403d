#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p
timeout 900 python -m pytest tests/test_reference_suite.py -m gpu -q -s > gpurun_out/r3p/suite.log 2>&1
grep -E "tests, [0-9]+ pass|passed|failed" gpurun_out/r3p/suite.log | head
PYTHONPATH=robotics-toolbox-python_amd python - <<'PY'
import numpy as np, rtbhip
from rtbhip import urdf
ur5 = urdf.load("UR5").ets()
Tep = ur5.eval([0, -0.3, 0, -2.2, 0, 2.0])
for pinv in (False, True):
    s = rtbhip.IK_GN(pinv=pinv, joint_limits=True, seed=0, tol=1e-6)
    sol = s.solve(ur5, Tep)
    print("pinv", pinv, sol, "E of q:", s.error(ur5.eval(sol.q), Tep)[1], "qlim ok:", bool(((sol.q >= ur5.qlim[0]) & (sol.q <= ur5.qlim[1])).all()))
print(ur5.qlim)
PY

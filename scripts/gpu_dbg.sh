#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/dbg3.py <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "robotics-toolbox-python_amd")]
import numpy as np, rtbhip
which, mode, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
arm = rtbhip.models.DH.Panda() if which == "panda" else rtbhip.models.DH.Puma560()
rng = np.random.default_rng(0)
q = rng.uniform(-1, 1, (N, arm.n)); qd = rng.normal(size=(N, arm.n)); tq = rng.normal(size=(N, arm.n))
if mode == "inertia": out = arm.inertia(q)
elif mode == "accel": out = arm.accel(q, qd, tq)
else: out = arm.coriolis(q, qd)
print(which, mode, N, "ok", np.asarray(out).shape, float(np.abs(out).max()), flush=True)
PY
for a in "panda inertia 1" "panda inertia 64" "panda inertia 1000" "puma inertia 64" "panda accel 64" "puma accel 64" "panda coriolis 64"; do timeout 60 python /tmp/dbg3.py $a 2>&1 | grep -v amdgpu.ids | tail -2; done

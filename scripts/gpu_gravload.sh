#!/bin/bash
# gravload through k_rne_atrest (qd = NULL) against the full recursion fed zeros; then the whole -m gpu suite
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/gl.py <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
rob = rtbhip.models.DH.Panda()
N = 1000000
rng = np.random.default_rng(6)
q = torch.from_numpy(rng.uniform(rob.qlim[0], rob.qlim[1], (N, 7))).cuda()
z = torch.zeros_like(q)
def t(fn, reps=40):
    for _ in range(5): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev: a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0]
for r in range(3):
    print("gravload (at rest)    avg %.4f min %.4f ms" % t(lambda: rob.gravload(q)))
    print("rne(q, zeros, zeros)  avg %.4f min %.4f ms" % t(lambda: rob.rne(q, z, z)))
PY
timeout 200 python /tmp/gl.py 2>&1 | grep -v amdgpu.ids
bash scripts/gpu_suite.sh

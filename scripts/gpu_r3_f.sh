#!/bin/bash
# Round 3, visit f: k_rne one tile per workgroup vs the persistent form (2 / 3 / 4 waves per SIMD), interleaved; parity of the persistent form.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python - <<'PY'
import numpy as np, torch, sys
sys.path[:0] = ['.', 'robotics-toolbox-python_amd']
import rtbhip
rob = rtbhip.models.DH.Panda()
rng = np.random.default_rng(3)
N = 1250000
q = torch.from_numpy(rng.uniform(rob.qlim[0], rob.qlim[1], (N, 7))).cuda(); qd = torch.from_numpy(rng.normal(size=(N, 7))).cuda(); qdd = torch.from_numpy(rng.normal(size=(N, 7))).cuda()
base = rob.rne(q, qd, qdd).clone()
for w in (1, 2, 3, 4):
    rtbhip.tune("rne_persist", w)
    t = rob.rne(q, qd, qdd)
    print("persist", w, "bit-equal to one-tile-per-workgroup:", bool((t == base).all()))
rtbhip.tune("rne_persist", 0)
PY
for rep in 1 2 3; do for t in 0 3 2 4; do
  timeout 300 python bench_extra.py --what rne --no-cpu --steps 60 --tune rne_persist=$t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rne_persist=$t', 'avg %.4f min %.4f ms' % (d['kernel_avg_ms'], d['kernel_min_ms']), '%.3f of HBM' % d['roofline']['frac'])"
done; done
for t in 0 3; do
  timeout 300 python bench_extra.py --what rne --no-cpu --steps 30 --n-rne 10000000 --tune rne_persist=$t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1e7 triples rne_persist=$t', 'avg %.4f min %.4f ms' % (d['kernel_avg_ms'], d['kernel_min_ms']), '%.3f of HBM' % d['roofline']['frac'])"
done

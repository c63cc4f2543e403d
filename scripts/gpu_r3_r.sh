#!/bin/bash
# Round 3, visit r: k_rne with 1 / 2 / 4 waves per workgroup (fewer workgroup launches for the same waves), interleaved; parity of the variants.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
PYTHONPATH=robotics-toolbox-python_amd python - <<'PY'
import numpy as np, torch, rtbhip
arm = rtbhip.models.DH.Panda()
rng = np.random.default_rng(0)
q, qd, qdd = (torch.from_numpy(rng.normal(size=(100003, 7))).cuda() for _ in range(3))
base = arm.rne(q, qd, qdd).cpu().numpy()
for w in (2, 4):
    rtbhip.tune("rne_wpb", w)
    assert np.array_equal(arm.rne(q, qd, qdd).cpu().numpy(), base), w
rtbhip.tune("rne_wpb", 1)
print("wpb variants bit-equal")
PY
for rep in 1 2 3; do
for w in 1 2 4; do
  timeout 300 python bench_extra.py --what rne --no-cpu --steps 30 --tune rne_wpb=$w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('wpb=$w', 'step %.4f kernel avg %.4f min %.4f ms' % (d['ms_per_step'], d['kernel_avg_ms'], d['kernel_min_ms']))"
done
done
timeout 300 python bench_extra.py --what rne --no-cpu --steps 10 --n-rne 10000000 --tune rne_wpb=1 2>/dev/null | cut -c1-200
timeout 300 python bench_extra.py --what rne --no-cpu --steps 10 --n-rne 10000000 --tune rne_wpb=4 2>/dev/null | cut -c1-200

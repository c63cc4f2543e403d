#!/bin/bash
# Round 4, visit y: the 13..16-joint instantiations of the tree dynamics kernels on the device -- tree tests and the dynamics fuzz.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4y}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_erobot_dynamics.py tests/test_erobot_rne.py tests/test_dynamics_terms.py -q -m gpu 2>&1 | tail -4 | tee $O/pytest_tree.log
timeout 600 python scripts/gpu_fuzz_dyn.py > $O/fuzz_dyn.jsonl 2> $O/fuzz_dyn.err; echo "fuzz rc=$?" >> $O/fuzz_dyn.jsonl; tail -4 $O/fuzz_dyn.jsonl | cut -c1-300
python - <<'PY' | tee $O/yumi_dyn.jsonl
import sys, os, json
sys.path[:0] = [os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"], "robotics-toolbox-python_amd")]
import numpy as np, torch
from rtbhip import urdf
from benchlib import sustained_ms
rob = urdf.load("YuMi"); arms = ("gripper_r_base", "gripper_l_base"); er = rob.erobot(arms)
N = 200000; rng = np.random.default_rng(1)
q, qd, tq = (torch.from_numpy(x).cuda() for x in (rng.uniform(-1.5, 1.5, (N, er.n)), rng.normal(size=(N, er.n)), rng.normal(size=(N, er.n))))
for term, fn in (("rne", lambda: er.rne(q, qd, tq)), ("inertia", lambda: er.inertia(q)), ("coriolis", lambda: er.coriolis(q, qd)), ("accel", lambda: er.accel(q, qd, tq))):
    fn(); ms, _, _ = sustained_ms(fn)
    print(json.dumps({"robot": "YuMi, both arms (14 joints)", "term": term, "N": N, "sustained_ms": round(ms, 4), "per_s": N / (ms * 1e-3)}), flush=True)
PY

#!/bin/bash
# Round 2, visit i: IK_QP and the robot-wide-q chains on the GPU; timing of ikine_QP
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2i
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_00_gpu_parity.py tests/test_03_python_ik_pins.py -m gpu -q --timeout 200 --tb=short -k "qp or robot_wide or ikine_equals" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
grep -v "Warning\|^  \|^$" $O/pytest_gpu.log | tail -25 | cut -c1-250
cat > /tmp/qp_time.py <<'PY'
import sys, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
ets = rtbhip.models.Panda().ets(); ets.qlim = rtbhip.models.PANDA_QLIM
N = 100000
rng = np.random.default_rng(1)
Tep = ets.eval(torch.from_numpy(rng.uniform(ets.qlim[0], ets.qlim[1], (N, 7))).cuda())
for name, fn in (("ikine_LM", lambda: ets.ikine_LM(Tep, seed=2)), ("ikine_QP kj=0.01 (IK_QP class default)", lambda: ets.ikine_QP(Tep, seed=2, kj=0.01)),
                 ("ikine_QP kj=1 (ETS.ikine_QP default)", lambda: ets.ikine_QP(Tep, seed=2)), ("ikine_QP kj=0.01 kq=1 (velocity-damper rows)", lambda: ets.ikine_QP(Tep, seed=2, kj=0.01, kq=1.0)), ("ikine_NR pinv", lambda: ets.ikine_NR(Tep, seed=2, pinv=True))):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); s = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%-46s %7.2f ms  success %.4f  mean it %.1f" % (name, dt * 1e3, s.each["success"].mean(), s.each["iterations"].mean()))
PY
timeout 200 python /tmp/qp_time.py 2>&1 | grep -v amdgpu.ids | tee $O/qp_time.txt

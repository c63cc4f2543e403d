#!/bin/bash
# Round 3, visit y: four builds of the IK kernels on one box, interleaved: r2 = the round-2 final tree; B = round 3 with the flat schedule and the
# counters as run-time switches (234 VGPRs); C = those as template parameters + the unit-weight copy of the LM step (256 VGPRs + 12 B scratch);
# D = template parameters, no unit-weight copy (223 VGPRs).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=$R/robotics-toolbox-python_amd/lib/variants
run() { # dir lib label
  (cd $1 && RTBHIP_LIB=$2 timeout 300 python bench_extra.py --what ik --no-cpu --steps 8 2>/dev/null) | python -c "
import sys,json
print('$3', ' | '.join('%.4f (min %.4f)' % (json.loads(l)['kernel_avg_ms'], json.loads(l)['kernel_min_ms']) for l in sys.stdin if l.startswith('{')))"
}
for rep in 1 2 3; do
  run $R/r2cmp "" "r2"
  run $R $V/ikB.so "B "
  run $R "" "C "
  run $R $V/ikD.so "D "
done
cd $R && RTBHIP_LIB=$V/ikD.so timeout 600 python -m pytest tests/test_00_gpu_parity.py tests/test_03_python_ik_pins.py -m gpu -q -x -k "ik or IK" 2>&1 | tail -2

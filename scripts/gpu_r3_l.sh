#!/bin/bash
# Round 3, visit l: k_ik at three waves per SIMD (168 VGPRs + 204 B scratch, 8-entry search ring so that 12 waves' LDS fit a CU) against the
# shipped two-wave build: parity tests through the variant, then interleaved timings at 1e5 / 1e6 targets.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3l
mkdir -p $O
cd $R
V=$R/robotics-toolbox-python_amd/lib/variants/ik3.so
RTBHIP_LIB=$V timeout 900 python -m pytest tests/test_00_gpu_parity.py tests/test_03_python_ik_pins.py -m gpu -q -x -k "ik or IK" --timeout 600 2>&1 | tail -4
for rep in 1 2 3; do
  timeout 300 python bench_extra.py --what ik --no-cpu --steps 8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('base      ', d['metric'][22:70], 'avg %.4f min %.4f' % (d['kernel_avg_ms'], d['kernel_min_ms']))"
  for w in 12 10; do
  RTBHIP_LIB=$V timeout 300 python bench_extra.py --what ik --no-cpu --steps 8 --tune ik_waves_per_cu=$w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('3-wave w=$w', d['metric'][22:70], 'avg %.4f min %.4f' % (d['kernel_avg_ms'], d['kernel_min_ms']))"
  done
done
RTBHIP_LIB=$V timeout 300 python bench_extra.py --what ik --no-cpu --steps 8 --tune ik_waves_per_cu=8 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('3-wave-build w=8', d['metric'][22:70], 'avg %.4f min %.4f' % (d['kernel_avg_ms'], d['kernel_min_ms']))"

#!/bin/bash
# A/B of the RNE variants under lib/variants: parity tests through each variant, then interleaved timing
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in robotics-toolbox-python_amd/lib/variants/*.so; do
  echo "== parity with $(basename $v)"
  RTBHIP_LIB=$GRAFT_REPO_ROOT/$v timeout 300 python -m pytest tests/test_00_gpu_parity.py tests/test_dynamics_terms.py -m gpu -q --timeout 200 -k "rne or dyn or gravload or G9" 2>&1 | grep -E "passed|failed" | tail -2
done
bash scripts/gpu_ab_rne2.sh

#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "" $R/robotics-toolbox-python_amd/lib/variants/pu3.so $R/robotics-toolbox-python_amd/lib/variants/pu4.so; do echo "== ${v:-U2}"; RTBHIP_LIB=$v ./scripts/gpu_p4.sh; done

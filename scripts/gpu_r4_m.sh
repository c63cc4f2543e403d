#!/bin/bash
# Round 4, visit m: pinned segment loads in the GENERAL walk too (variant ik_pin2) -- measured on the general-walk kernels (ik_plain = 0, and a
# weighted mask) of both libraries.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4m}
mkdir -p $O
cd $R
V=$R/robotics-toolbox-python_amd/lib/variants
for round in 1 2 3; do
  for lib in "" $V/ik_pin2.so; do
    RTBHIP_LIB=$lib IK_AB_SET=7 timeout 300 python scripts/ik_ab.py 100000 1 2>/dev/null | sed "s#^#$(basename ${lib:-product}) #" | cut -c1-200
  done
done

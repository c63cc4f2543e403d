#!/bin/bash
# build_ik_variant.sh NAME -DFOO=1 ... : robotics-toolbox-python_amd/lib/variants/NAME.so = the product's objects (build/obj, from build_lib) with
# ik_kernels.hip recompiled under the extra defines -- an A/B library for RTBHIP_LIB in ~2 minutes instead of a full rebuild.
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
mkdir -p $R/robotics-toolbox-python_amd/lib/variants $R/build/variant
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -mllvm -pragma-unroll-threshold=1048576 -I$R/include "$@" -c $R/robotics-toolbox-python_amd/csrc/ik_kernels.hip -o $R/build/variant/ik_$name.o
objs=$(ls $R/build/obj/*.o | grep -v ik_kernels.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build/variant/ik_$name.o -o $R/robotics-toolbox-python_amd/lib/variants/$name.so
ls -la $R/robotics-toolbox-python_amd/lib/variants/$name.so

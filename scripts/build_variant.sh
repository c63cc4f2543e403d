#!/bin/bash
# build_variant.sh SRC NAME -DFOO=1 ... : robotics-toolbox-python_amd/lib/variants/NAME.so = the product's objects (build/obj, from build_lib) with
# csrc/SRC.hip recompiled under the extra defines -- an A/B library for RTBHIP_LIB without a full rebuild (build_ik_variant.sh for any kernel file).
set -e
R=$(cd $(dirname $0)/.. && pwd)
src=$1; name=$2; shift 2
mkdir -p $R/robotics-toolbox-python_amd/lib/variants $R/build/variant
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -mllvm -pragma-unroll-threshold=1048576 -I$R/include "$@" -c $R/robotics-toolbox-python_amd/csrc/$src.hip -o $R/build/variant/${src}_$name.o
objs=$(ls $R/build/obj/*.o | grep -v $src.hip.o)
hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build/variant/${src}_$name.o -o $R/robotics-toolbox-python_amd/lib/variants/$name.so
ls -la $R/robotics-toolbox-python_amd/lib/variants/$name.so

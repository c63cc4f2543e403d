#!/bin/bash
# build_variant.sh NAME -DFOO=1 ... : builds robotics-toolbox-python_amd/lib/variants/NAME.so with extra defines (A/B runs via RTBHIP_LIB)
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
mkdir -p $R/robotics-toolbox-python_amd/lib/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w -I$R/include "$@" $R/robotics-toolbox-python_amd/csrc/*.cpp $R/robotics-toolbox-python_amd/csrc/*.hip -o $R/robotics-toolbox-python_amd/lib/variants/$name.so

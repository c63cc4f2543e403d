#!/bin/bash
# A/B of compiled kernel variants (robotics-toolbox-python_amd/lib/variants/*.so) + practical HBM ceiling probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
./scripts/roofline_probe.bin > gpurun_out/roofline_probe.txt 2>&1; cat gpurun_out/roofline_probe.txt
run() { # name lib n
  RTBHIP_LIB=$2 python bench.py --steps 60 --warmup 5 --no-cpu --n $3 > gpurun_out/var_$1_$3.json 2> gpurun_out/var_$1_$3.err
  echo "$1 n=$3: $(cat gpurun_out/var_$1_$3.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("%.4g cfg/s  avg %.4f ms  min %.4f ms  frac %.3f" % (d["value"], d["roofline"]["kernel_avg_ms"], d["roofline"]["kernel_min_ms"], d["roofline"]["frac"]))')"
}
for n in 1000000 4000000; do
  run default "" $n
  for v in $R/robotics-toolbox-python_amd/lib/variants/*.so; do run $(basename $v .so) $v $n; done
done

#!/bin/bash
# Round 2, visit b: the tests that failed in visit a, the host-pointer pipeline, host-overhead diagnosis, headline with host_path
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2b
mkdir -p $O
timeout 1500 python -m pytest tests/test_02_compat_shim.py tests/test_dist_gpu.py tests/test_host_path.py tests/test_00_gpu_parity.py tests/test_hip_graph.py -m gpu -q --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 300 python scripts/diag_host_overhead.py 2>&1 | tail -30
python bench.py --steps 200 --warmup 20 > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-400 $O/bench_n1.json; python -c "
import json; d=json.load(open('$O/bench_n1.json')); print(json.dumps(d.get('host_path'), indent=1))"; tail -3 $O/bench_n1.err
for mb in 8 16 64; do RTBHIP_HOST_CHUNK_MB=$mb python -c "
import sys; sys.path[:0]=['.','robotics-toolbox-python_amd']
import numpy as np, time, rtbhip
ets=rtbhip.models.Panda().ets(); q=np.random.default_rng(0).uniform(-3,3,(1000000,7))
ets.fkine_jacob0(q[:5000]); best=9
for _ in range(4):
    t=time.perf_counter(); o=ets.fkine_jacob0(q); best=min(best,time.perf_counter()-t)
Ta,Ja=np.empty((1000000,4,4)),np.empty((1000000,6,7))
from rtbhip._lib import lib,host_ptr,check
b2=9
for _ in range(3):
    t=time.perf_counter(); check(lib().rtbhip_fkine_jacob(ets._handle(),host_ptr(q),1000000,None,None,0,host_ptr(Ta),host_ptr(Ja),0,None)); b2=min(b2,time.perf_counter()-t)
print('chunk %s MB: pinned results %.2f ms (%.1f GB/s)   pageable results %.2f ms (%.1f GB/s)'%('$mb',best*1e3,0.52/best,b2*1e3,0.52/b2))
"; done

cd $GRAFT_REPO_ROOT; O=gpurun_out/r6e; mkdir -p $O; export TMPDIR=/tmp
run() { echo "=== $*" >> $O/diag.log; timeout 600 python scripts/jit_diag.py "$@" >> $O/diag.log 2>&1; echo "rc=$?" >> $O/diag.log; }
run tree YuMi; run tree Panda
grep -v amdgpu.ids $O/diag.log | cut -c1-900
timeout 2400 python -m pytest tests/test_jit_gpu.py tests/test_large_chains_gpu.py -m gpu -q -rf --timeout 1200 > $O/pytest_new.log 2>&1; tail -40 $O/pytest_new.log | cut -c1-250

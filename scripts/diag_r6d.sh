cd $GRAFT_REPO_ROOT; O=gpurun_out/r6d; mkdir -p $O; export TMPDIR=/tmp
run() { echo "=== $*" >> $O/diag.log; timeout 600 python scripts/jit_diag.py "$@" >> $O/diag.log 2>&1; echo "rc=$?" >> $O/diag.log; }
run rne; run ik Puma560; run ik Mico; run tree AL5D; run tree LBR; run tree KinovaGen3; run chain 13; run chain 18 3
run tree YuMi rne; run tree YuMi gravload; run tree YuMi inertia
grep -v amdgpu.ids $O/diag.log | cut -c1-700
timeout 1500 python -m pytest tests/test_jit_gpu.py tests/test_large_chains_gpu.py -m gpu -q -rf --timeout 900 -x --deselect "tests/test_jit_gpu.py::test_link_trees_without_builtin_instantiation[YuMi]" > $O/pytest_new.log 2>&1; tail -30 $O/pytest_new.log | cut -c1-300

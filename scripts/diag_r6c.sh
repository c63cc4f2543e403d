cd $GRAFT_REPO_ROOT; O=gpurun_out/r6c; mkdir -p $O; export TMPDIR=/tmp
run() { echo "=== $*" >> $O/diag.log; timeout 300 python scripts/jit_diag.py "$@" >> $O/diag.log 2>&1; echo "rc=$?" >> $O/diag.log; }
for n in 13 15 16 17 18 20 24; do run chain $n; done
run chain 12 3; run chain 18 3
cat $O/diag.log | grep -v amdgpu.ids | cut -c1-400

cd $GRAFT_REPO_ROOT; O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
run() { echo "=== $*" >> $O/diag.log; timeout 300 python scripts/jit_diag.py "$@" >> $O/diag.log 2>&1; echo "rc=$?" >> $O/diag.log; }
run rne
RTBHIP_RNE_SIG_AND=ffffffffffffffe7 run rne      # link 0 alpha class off
for m in e7f3f9fcfe7f3fe7 9fcfe7f3f9fcfe7f 1fffffffffffffff fc7e3f1f8fc7e3f8; do RTBHIP_RNE_SIG_AND=$m run rne $m; done
run ik Puma560; run ik LBR; run ik px100
run tree AL5D rne; run tree AL5D; run tree Puma560; run tree LBR; run tree KinovaGen3 rne; run tree KinovaGen3; run tree YuMi rne
cat $O/diag.log | cut -c1-900

#!/bin/bash
# Round 4, visit f: third IK knob sweep (first chunk around 8 searches, fresh share around 100 %), candidates at other sizes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4f}
mkdir -p $O
cd $R
IK_AB_SET=4 timeout 900 python scripts/ik_ab.py 100000 2 > $O/ik_ab4.jsonl 2> $O/ik_ab4.err; cut -c1-200 $O/ik_ab4.jsonl; tail -2 $O/ik_ab4.err
for n in 10000 20000 50000 150000 200000 300000 380000; do
  echo "N=$n"; IK_AB_SET=5 timeout 600 python scripts/ik_ab.py $n 2 > $O/ik_ab5_$n.jsonl 2>> $O/ik_ab5.err; cut -c1-200 $O/ik_ab5_$n.jsonl
done
echo "notebook 1e5"; IK_AB_SET=5 IK_AB_NOTEBOOK=1 timeout 600 python scripts/ik_ab.py 100000 2 > $O/ik_ab5_notebook.jsonl 2>> $O/ik_ab5.err; cut -c1-200 $O/ik_ab5_notebook.jsonl

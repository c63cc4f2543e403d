#!/bin/bash
# Round 4, visit e: second IK knob sweep (around fresh 100 %), then the candidates at other batch sizes and in the notebook setting.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4e}
mkdir -p $O
cd $R
IK_AB_SET=2 timeout 900 python scripts/ik_ab.py 100000 2 > $O/ik_ab2.jsonl 2> $O/ik_ab2.err; cut -c1-200 $O/ik_ab2.jsonl; tail -2 $O/ik_ab2.err
for n in 20000 50000 200000 400000 1000000; do
  echo "N=$n"; IK_AB_SET=3 timeout 600 python scripts/ik_ab.py $n 2 > $O/ik_ab3_$n.jsonl 2>> $O/ik_ab3.err; cut -c1-200 $O/ik_ab3_$n.jsonl
done
echo "notebook 1e5"; IK_AB_SET=3 IK_AB_NOTEBOOK=1 timeout 600 python scripts/ik_ab.py 100000 2 > $O/ik_ab3_notebook.jsonl 2>> $O/ik_ab3.err; cut -c1-200 $O/ik_ab3_notebook.jsonl

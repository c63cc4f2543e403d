#!/bin/bash
# Round 4, visit t: k_tree_dyn's unit-acceleration passes -- the full recursion (variant tree_full), acceleration-only (tree_acc), acceleration-only
# from the column's own group on (the product) -- interleaved, sustained, with the tree dynamics parity tests on the product.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${VISIT:-r4t}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_erobot_dynamics.py tests/test_erobot_rne.py -q -m gpu 2>&1 | tail -3 | tee $O/pytest_tree.log
V=$R/robotics-toolbox-python_amd/lib/variants
for round in 1 2; do
  for lib in "" $V/tree_full.so $V/tree_acc.so; do
    RTBHIP_LIB=$lib TREE_AB_TAG=$(basename ${lib:-product}) timeout 300 python scripts/tree_ab.py 2>/dev/null | tee -a $O/tree_ab.jsonl
  done
done

#!/bin/bash
export EXTRA_ARGS="--n-ik 1000000"
$GRAFT_REPO_ROOT/scripts/gpu_pmc_sq.sh ik 2>&1 | grep "k_ik"

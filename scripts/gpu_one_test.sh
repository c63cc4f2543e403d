#!/bin/bash
# run a subset of the GPU suite: bash scripts/gpu_one_test.sh <pytest args>
cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest -m gpu -q -rf "$@" 2>&1 | grep -vE "^\s*$" | tail -40

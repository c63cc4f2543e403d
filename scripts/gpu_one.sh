#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dynamics_terms.py -m gpu -q --timeout 600 --tb=short 2>&1 | grep -v "Warning\|^  \|^$" | tail -12 | cut -c1-300

#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 --tb=short -k "row_blocks or ik" 2>&1 | grep -v "Warning\|^  \|^$" | tail -15 | cut -c1-300

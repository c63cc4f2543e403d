cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_00_gpu_parity.py -m gpu -q -x -k "specialisations" 2>&1 | grep -vE "^\s*$" | tail -40

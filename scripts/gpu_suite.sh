#!/bin/bash
# the whole -m gpu suite + smoke, nothing else
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -6
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

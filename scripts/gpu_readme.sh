#!/bin/bash
# runs the README's "Use" snippet on the GPU box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
sed -n '/^```python/,/^```$/p' README.md | sed '1d;$d' > /tmp/readme_snippet.py
PYTHONPATH=$GRAFT_REPO_ROOT/robotics-toolbox-python_amd timeout 300 python /tmp/readme_snippet.py 2>&1 | grep -v amdgpu.ids | tail -5; echo "rc=${PIPESTATUS[0]}"

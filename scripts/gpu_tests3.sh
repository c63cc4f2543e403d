#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -v Warning | tail -1; done

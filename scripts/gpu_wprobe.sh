#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; ./scripts/write_probe.bin | tee gpurun_out/write_probe.txt

#!/bin/bash
# Round 4, first visit: the state the last (GPU-less) session of round 3 left -- the whole -m gpu suite (tests/test_p_servo.py and the reference's
# test_tools / test_Link / test_ELink run on a GPU for the first time here), smoke, headline bench, every secondary leg (the new `servo` lines
# included), rocprofv3 kernel stats and the PMC passes: scripts/gpu_r3_g.sh under another name.  Results: gpurun_out/r4a.
VISIT=r4a bash $GRAFT_REPO_ROOT/scripts/gpu_r3_g.sh

#!/usr/bin/env python3
"""scripts/headline_placement_probe.py -- the headline kernel on K separately allocated (q, T, J) buffer sets in ONE process (all kept alive, so every set
has its own physical pages): the distribution of the sustained kernel time over placements, and over mixed sets (T of one, J of another)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
N, K = 1000000, int(os.environ.get("PROBE_K", 10))
ets = rtbhip.models.Panda().ets()
lib = rtbhip.lib(); h = ets._handle(); ets.upload()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
qh = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7)))
sets = []
for k in range(K):
    sets.append((qh.cuda(), torch.empty((N, 4, 4), dtype=torch.float64, device="cuda"), torch.empty((N, 6, 7), dtype=torch.float64, device="cuda")))
def t(q, T, J):
    p = [C.c_void_p(x.data_ptr()) for x in (q, T, J)]
    def f():
        assert lib.rtbhip_fkine_jacob(h, p[0], N, None, None, 0, p[1], p[2], 1, stream) == 0
    f(); ms, _, _ = sustained_ms(f); return round(ms * 1e3, 2)
own = [t(*s) for s in sets]
again = [t(*s) for s in sets]
mixed = [t(sets[0][0], sets[i][1], sets[(i + 1) % K][2]) for i in range(K)]
print(json.dumps({"own": own, "again": again, "mixed T_i J_i+1": mixed, "addr_T": [hex(s[1].data_ptr() >> 21) for s in sets], "addr_J": [hex(s[2].data_ptr() >> 21) for s in sets]}))

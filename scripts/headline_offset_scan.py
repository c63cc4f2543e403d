#!/usr/bin/env python3
"""scripts/headline_offset_scan.py -- q, T and J carved out of ONE 4 GiB allocation; T fixed, the Jacobian's offset swept in 2 MiB steps (and a few finer
ones): is there a relative offset at which the round-3 form (both arrays non-temporal) is fast, i.e. could one allocation with a known-good layout
replace the allocator's lottery?  Run with RTBHIP_LIB = a build with RTB_T_PLAIN_MAX_BYTES=0."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
N = int(os.environ.get("PROBE_N", 1000000))
ets = rtbhip.models.Panda().ets()
lib = rtbhip.lib(); h = ets._handle(); ets.upload()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
pool = torch.empty(4 * 1024 ** 3, dtype=torch.uint8, device="cuda")
MB = 1 << 20
q = pool[0:56 * N].view(torch.float64).view(N, 7); q.copy_(torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7))))
offT = 64 * MB
def t(offJ):
    T = pool[offT:offT + 128 * N].view(torch.float64); J = pool[offJ:offJ + 336 * N].view(torch.float64)
    p = [C.c_void_p(x.data_ptr()) for x in (q, T, J)]
    def f():
        assert lib.rtbhip_fkine_jacob(h, p[0], N, None, None, 0, p[1], p[2], 1, stream) == 0
    f(); ms, _, _ = sustained_ms(f, 0.01, 0.01) if False else sustained_ms(f); return round(ms * 1e3, 1)
out = {}
base = 256 * MB
for k in list(range(0, 64)) + [64, 96, 128, 192, 256, 384, 512, 768, 1024, 1536]:
    out["+%d MiB" % (2 * k)] = t(base + 2 * k * MB)
for fine in (4096, 65536, 262144, 1048576):
    out["+%d B" % fine] = t(base + fine)
print(json.dumps({"N": N, "us_by_J_offset": out}))

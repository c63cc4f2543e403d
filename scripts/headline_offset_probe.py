#!/usr/bin/env python3
"""scripts/headline_offset_probe.py -- does the headline kernel's time depend on WHERE its buffers lie?  One process, one big allocation; q, T and J are
carved out of it at controlled offsets (and, for comparison, allocated separately by torch as bench.py does), the kernel is timed sustained for each
placement, several rounds interleaved.  Leases of this round showed the same sources at 81 and at 92 us on one box in two consecutive processes."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
N = 1000000
ets = rtbhip.models.Panda().ets()
lib = rtbhip.lib(); h = ets._handle(); ets.upload()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
qh = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7)))
pool = torch.empty(2 * 1024 ** 3, dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
MB = 1 << 20
def place(off_q, off_T, off_J):
    q = pool[off_q:off_q + 56 * N].view(torch.float64).view(N, 7); q.copy_(qh)
    T = pool[off_T:off_T + 128 * N].view(torch.float64); J = pool[off_J:off_J + 336 * N].view(torch.float64)
    return q, T, J
cases = {"torch.empty x3": None}
for name, (a, b, c) in {"packed q|T|J": (0, 56 * N, 56 * N + 128 * N), "2 MiB aligned": (0, 64 * MB, 256 * MB), "2 MiB + 4 KiB stagger": (0, 64 * MB + 4096, 256 * MB + 8192),
                        "2 MiB + 64 KiB stagger": (0, 64 * MB + 65536, 256 * MB + 131072), "2 MiB + 1 MiB stagger": (0, 64 * MB + MB, 256 * MB + MB // 2),
                        "1 GiB apart": (0, 512 * MB, 1024 * MB), "odd 256 B offsets": (256, 64 * MB + 768, 256 * MB + 1280)}.items():
    cases[name] = (a, b, c)
res = {k: [] for k in cases}
for rnd in range(3):
    for name, offs in cases.items():
        if offs is None:
            q = qh.cuda(); T = torch.empty((N, 4, 4), dtype=torch.float64, device="cuda"); J = torch.empty((N, 6, 7), dtype=torch.float64, device="cuda")
        else:
            q, T, J = place(*offs)
        p = [C.c_void_p(x.data_ptr()) for x in (q, T, J)]
        def f():
            assert lib.rtbhip_fkine_jacob(h, p[0], N, None, None, 0, p[1], p[2], 1, stream) == 0
        f(); ms, _, _ = sustained_ms(f)
        res[name].append(round(ms * 1e3, 2))
        if offs is None:
            res.setdefault("torch.empty x3 addresses", []).append([hex(x.data_ptr() & 0xffffffffff) for x in (q, T, J)])
for k, v in res.items():
    print(json.dumps({"placement": k, "us": v}), flush=True)

#!/usr/bin/env python3
"""scripts/headline_alloc_probe.py MODE -- the headline kernel on buffers allocated (a) first thing in the process, as bench.py does, (b) after a 2 GiB
allocation that stays, (c) after a 2 GiB allocation that was released (torch cache emptied), (d) through hipMalloc directly (no torch cache), (e) after
many small allocations.  One mode per process; prints the buffers' addresses and the sustained kernel time."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
mode = sys.argv[1]
N = 1000000
torch.cuda.init(); torch.zeros(1, device="cuda")
keep = []
if mode == "b": keep.append(torch.empty(2 * 1024 ** 3, dtype=torch.uint8, device="cuda"))
if mode == "c":
    x = torch.empty(2 * 1024 ** 3, dtype=torch.uint8, device="cuda"); del x; torch.cuda.empty_cache()
if mode == "e": keep += [torch.empty(3 * 1024 * 1024 + 4096 * i, dtype=torch.uint8, device="cuda") for i in range(64)]
if mode == "f": keep.append(torch.empty(8 * 1024 ** 3, dtype=torch.uint8, device="cuda"))
if mode == "g": keep.append(torch.empty(256 * 1024 ** 2, dtype=torch.uint8, device="cuda"))
ets = rtbhip.models.Panda().ets()
lib = rtbhip.lib(); h = ets._handle(); ets.upload()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
qh = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7)))
q = qh.cuda(); T = torch.empty((N, 4, 4), dtype=torch.float64, device="cuda"); J = torch.empty((N, 6, 7), dtype=torch.float64, device="cuda")
p = [C.c_void_p(x.data_ptr()) for x in (q, T, J)]
def f():
    assert lib.rtbhip_fkine_jacob(h, p[0], N, None, None, 0, p[1], p[2], 1, stream) == 0
f(); out = []
for _ in range(3):
    ms, _, _ = sustained_ms(f); out.append(round(ms * 1e3, 2))
print(json.dumps({"mode": mode, "us": out, "q": hex(q.data_ptr()), "T": hex(T.data_ptr()), "J": hex(J.data_ptr())}), flush=True)

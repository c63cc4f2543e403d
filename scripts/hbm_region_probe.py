#!/usr/bin/env python3
"""scripts/hbm_region_probe.py -- K separately allocated 336 MB buffers in one process: plain write-only (fill) and read-only (sum) bandwidth of EACH, and the
headline kernel with each as its J output (one fixed q and T).  Is the 78 / 90 us split of the headline kernel a property of the memory a buffer landed on?"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
N, K = 1000000, int(os.environ.get("PROBE_K", 16))
ets = rtbhip.models.Panda().ets()
lib = rtbhip.lib(); h = ets._handle(); ets.upload()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
q = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7))).cuda()
Ts = [torch.empty((N, 4, 4), dtype=torch.float64, device="cuda") for _ in range(4)]
Js = [torch.empty((N, 6, 7), dtype=torch.float64, device="cuda") for _ in range(K)]
def kern(T, J):
    p = [C.c_void_p(x.data_ptr()) for x in (q, T, J)]
    def f():
        assert lib.rtbhip_fkine_jacob(h, p[0], N, None, None, 0, p[1], p[2], 1, stream) == 0
    f(); ms, _, _ = sustained_ms(f); return round(ms * 1e3, 1)
rows = []
for k, J in enumerate(Js):
    w = sustained_ms(lambda: J.fill_(1.0))[0]; r = sustained_ms(lambda: J.sum())[0]
    rows.append({"J": k, "fill_GBs": round(336e6 / (w * 1e-3) / 1e9), "sum_GBs": round(336e6 / (r * 1e-3) / 1e9), "kernel_us_by_T": [kern(T, J) for T in Ts]})
for r in rows: print(json.dumps(r), flush=True)

#!/usr/bin/env python3
"""scripts/rne_placement_probe.py -- k_rne (DH Panda) on K separately allocated (q, qd, qdd, tau) buffer sets in one process: does its time depend on the
placement of its four arrays, as the headline kernel's did (profiles/r04_headline_stores.txt)?"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
rob = rtbhip.models.DH.Panda()
lib = rtbhip.lib(); dh = rob._dyn_handle()
grav = np.ascontiguousarray(rob._gravity_c(None)); gp = grav.ctypes.data_as(C.c_void_p)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for N, K in ((1250000, 10), (10000000, 5)):
    g = torch.Generator(device="cuda").manual_seed(3)
    times = []
    keep = []
    for k in range(K):
        q = torch.rand((N, 7), dtype=torch.float64, device="cuda", generator=g) * 2 - 1
        qd = torch.randn((N, 7), dtype=torch.float64, device="cuda", generator=g); qdd = torch.randn((N, 7), dtype=torch.float64, device="cuda", generator=g)
        tau = torch.empty((N, 7), dtype=torch.float64, device="cuda")
        keep.append((q, qd, qdd, tau))
        p = [C.c_void_p(x.data_ptr()) for x in (q, qd, qdd, tau)]
        def f():
            assert lib.rtbhip_rne(dh, p[0], p[1], p[2], N, gp, None, p[3], 1, stream) == 0
        f(); ms, _, _ = sustained_ms(f); times.append(round(ms, 4))
    print(json.dumps({"N": N, "ms_per_buffer_set": times, "frac_hbm": [round(224.0 * N / (t * 1e-3) / 8e12, 3) for t in times]}), flush=True)
    del keep

#!/usr/bin/env python3
"""scripts/rne_ab.py -- k_rne's A/B knobs (rtbhip_tune) under SUSTAINED timing at the config-4 sizes: round 3 measured them on bursts."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
rob = rtbhip.models.DH.Panda()
lib = rtbhip.lib(); dh = rob._dyn_handle()
grav = np.ascontiguousarray(rob._gravity_c(None)); gp = grav.ctypes.data_as(C.c_void_p)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ql = torch.from_numpy(np.asarray(rob.qlim)).cuda()
VARIANTS = [("one tile per workgroup (shipped)", {}), ("persistent, 2 waves/SIMD", {"rne_persist": 2}), ("persistent, 3 waves/SIMD", {"rne_persist": 3}),
            ("persistent, 4 waves/SIMD", {"rne_persist": 4}), ("2 waves per workgroup", {"rne_wpb": 2}), ("4 waves per workgroup", {"rne_wpb": 4})]
for N in (10000000, 1250000):
    g = torch.Generator(device="cuda").manual_seed(3)
    q = ql[0] + (ql[1] - ql[0]) * torch.rand((N, 7), dtype=torch.float64, device="cuda", generator=g)
    qd = torch.randn((N, 7), dtype=torch.float64, device="cuda", generator=g); qdd = torch.randn((N, 7), dtype=torch.float64, device="cuda", generator=g)
    tau = torch.empty((N, 7), dtype=torch.float64, device="cuda")
    p = [C.c_void_p(x.data_ptr()) for x in (q, qd, qdd, tau)]
    def f():
        assert lib.rtbhip_rne(dh, p[0], p[1], p[2], N, gp, None, p[3], 1, stream) == 0
    base = None
    res = {n: [] for n, _ in VARIANTS}
    for r in range(2):
        for name, kv in VARIANTS:
            rtbhip.tune("rne_persist", 0); rtbhip.tune("rne_wpb", 1)
            for k, v in kv.items():
                rtbhip.tune(k, v)
            f(); ms, _, _ = sustained_ms(f)
            t = tau.clone()
            if base is None: base = t
            res[name].append((round(ms, 5), bool(torch.equal(t, base))))
    rtbhip.tune("rne_persist", 0); rtbhip.tune("rne_wpb", 1)
    for name, _ in VARIANTS:
        print(json.dumps({"n": N, "variant": name, "sustained_ms": [m for m, _ in res[name]], "bit_equal": all(e for _, e in res[name]), "frac_hbm": 224.0 * N / (min(m for m, _ in res[name]) * 1e-3) / 8e12}), flush=True)
    del q, qd, qdd, tau

#!/usr/bin/env python3
"""scripts/headline_contig_probe.py -- the headline kernel on buffers from hipMalloc (default) and from hipExtMallocWithFlags(hipDeviceMallocContiguous),
K sets each in one process, all kept alive: is the slow / fast split of a placement (90 / 78 us) a matter of physical contiguity?"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
N, K = 1000000, int(os.environ.get("PROBE_K", 8))
ets = rtbhip.models.Panda().ets()
lib = rtbhip.lib(); h = ets._handle(); ets.upload()
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
hip = None
for m in open("/proc/self/maps"):
    if "libamdhip64" in m:
        hip = C.CDLL(m.split()[-1]); break
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
qh = np.ascontiguousarray(np.random.default_rng(0).uniform(-3, 3, (N, 7)))
def alloc(nbytes, flag):
    p = C.c_void_p()
    rc = hip.hipMalloc(C.byref(p), nbytes) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), nbytes, flag)
    assert rc == 0, (rc, flag)
    return p
out = {}
for label, flag in (("hipMalloc", None), ("contiguous", 0x4), ("uncached", 0x3)):
    times = []
    for k in range(K):
        try:
            q, T, J = alloc(56 * N, flag), alloc(128 * N, flag), alloc(336 * N, flag)
        except AssertionError as e:
            times.append("alloc failed %s" % (e,)); break
        assert hip.hipMemcpy(q, qh.ctypes.data_as(C.c_void_p), 56 * N, 1) == 0
        def f():
            assert lib.rtbhip_fkine_jacob(h, q, N, None, None, 0, T, J, 1, stream) == 0
        f(); ms, _, _ = sustained_ms(f); times.append(round(ms * 1e3, 2))
    out[label] = times
print(json.dumps(out))

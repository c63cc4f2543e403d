#!/usr/bin/env python3
"""scripts/headline_remap_probe.py -- the headline kernel of TWO libraries (the product, and a variant of kin_kernels.hip) on the SAME buffers in one
process: the process-to-process spread of this kernel (77-93 us: where the allocator put the buffers) is taken out of the comparison."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from rtbhip import _lib
from benchlib import sustained_ms
N = int(os.environ.get("PROBE_N", 1000000))
ets = rtbhip.models.Panda().ets()
libs = {"product": rtbhip.lib()}
for path in sys.argv[1:]:
    L = C.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        f = getattr(L, name); f.restype = res; f.argtypes = args
    libs[os.path.basename(path)] = L
handles = {}
rows = ets.optable()
arr = (_lib.rtbhip_et * len(rows))()
for i, (kind, flip, jindex, T) in enumerate(rows):
    arr[i].kind, arr[i].flip, arr[i].jindex = kind, flip, jindex
    arr[i].T[:] = list(np.asarray(T, dtype=np.float64).reshape(16))
for name, L in libs.items():
    h = C.c_uint64(0)
    assert L.rtbhip_chain_create(arr, len(rows), None, C.byref(h)) == 0
    assert L.rtbhip_chain_upload(h.value, -1) == 0
    handles[name] = h.value
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
q = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7))).cuda()
K = int(os.environ.get("PROBE_K", 6))
sets = [(torch.empty((N, 4, 4), dtype=torch.float64, device="cuda"), torch.empty((N, 6, 7), dtype=torch.float64, device="cuda")) for _ in range(K)]
res = {k: [] for k in libs}
ref = None
for T, J in sets:                              # one row of the result per buffer set: every library on the same buffers
    p = [C.c_void_p(x.data_ptr()) for x in (q, T, J)]
    for name, L in libs.items():
        def f():
            assert L.rtbhip_fkine_jacob(handles[name], p[0], N, None, None, 0, p[1], p[2], 1, stream) == 0
        f(); ms, _, _ = sustained_ms(f); res[name].append(round(ms * 1e3, 1))
        torch.cuda.synchronize()
        chk = float(T.sum() + J.sum())
        ref = chk if ref is None else ref
        assert chk == ref
for name in res: print(json.dumps({"lib": name, "us_per_buffer_set": res[name]}), flush=True)

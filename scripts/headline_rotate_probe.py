#!/usr/bin/env python3
"""scripts/headline_rotate_probe.py -- the headline kernel launched round-robin over R separately allocated output sets (R = 1: the bench's one set, rewritten
in place every step; R = 2, 3, 4: every launch writes arrays that were not the previous launch's), two libraries on the same buffers: how much of the
ordinary-store pose tile's 78 us rests on the pose array staying in the memory-side cache between steps?"""
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
rows = ets.optable()
arr = (_lib.rtbhip_et * len(rows))()
for i, (kind, flip, jindex, T) in enumerate(rows):
    arr[i].kind, arr[i].flip, arr[i].jindex = kind, flip, jindex
    arr[i].T[:] = list(np.asarray(T, dtype=np.float64).reshape(16))
handles = {}
for name, L in libs.items():
    h = C.c_uint64(0)
    assert L.rtbhip_chain_create(arr, len(rows), None, C.byref(h)) == 0 and L.rtbhip_chain_upload(h.value, -1) == 0
    handles[name] = h.value
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
q = torch.from_numpy(np.random.default_rng(0).uniform(-3, 3, (N, 7))).cuda()
sets = [(torch.empty((N, 4, 4), dtype=torch.float64, device="cuda"), torch.empty((N, 6, 7), dtype=torch.float64, device="cuda")) for _ in range(4)]
ptrs = [(C.c_void_p(T.data_ptr()), C.c_void_p(J.data_ptr())) for T, J in sets]
qp = C.c_void_p(q.data_ptr())
for R in (1, 2, 3, 4):
    for name, L in libs.items():
        state = {"k": 0}
        def f():
            Tp, Jp = ptrs[state["k"] % R]; state["k"] += 1
            assert L.rtbhip_fkine_jacob(handles[name], qp, N, None, None, 0, Tp, Jp, 1, stream) == 0
        f(); ms, _, _ = sustained_ms(f)
        print(json.dumps({"N": N, "rotating_sets": R, "lib": name, "us_per_launch": round(ms * 1e3, 1)}), flush=True)

#!/usr/bin/env python3
"""scripts/rne_1e7_probe.py -- k_rne at the full BASELINE configs[3] size (1e7 triples, 2.24 GB per launch): where does the spread between the
average and the fastest launch come from?  (Round-3 review: avg 0.59 ms against min 0.44 ms on one box.)

Three measurements, all through the C ABI on buffers allocated once unless stated:
  A  200 back-to-back launches, one HIP-event pair each: min / median / p90 / max, first 10 listed -- the steady state;
  B  the same with FRESH input and output buffers for every launch (torch.empty + fill: first touch of 2.24 GB of pages each time) -- the
     page-table / TLB hypothesis;
  C  a 4 s loop of launches with `rocm-smi --showclocks --showpower` sampled beside it -- the clock / power hypothesis;
  D  the 1.25e6-triple shard for comparison (same loop as A).
Prints one JSON object."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np
import torch
import rtbhip


def stats(ms):
    s = sorted(ms)
    n = len(s)
    return {"n": n, "min": s[0], "median": s[n // 2], "p90": s[int(0.9 * n)], "max": s[-1], "mean": sum(s) / n, "first10": [round(x, 4) for x in ms[:10]]}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
    rob = rtbhip.models.DH.Panda()
    lib = rtbhip.lib()
    dh = rob._dyn_handle()
    grav = np.ascontiguousarray(rob._gravity_c(None))
    gp = grav.ctypes.data_as(C.c_void_p)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ql = torch.from_numpy(np.asarray(rob.qlim)).cuda()

    def make(n):
        g = torch.Generator(device="cuda").manual_seed(3)
        q = ql[0] + (ql[1] - ql[0]) * torch.rand((n, 7), dtype=torch.float64, device="cuda", generator=g)
        qd = torch.randn((n, 7), dtype=torch.float64, device="cuda", generator=g)
        qdd = torch.randn((n, 7), dtype=torch.float64, device="cuda", generator=g)
        tau = torch.empty((n, 7), dtype=torch.float64, device="cuda")
        return q, qd, qdd, tau

    def launcher(bufs, n):
        p = [C.c_void_p(x.data_ptr()) for x in bufs]
        def f():
            rc = lib.rtbhip_rne(dh, p[0], p[1], p[2], n, gp, None, p[3], 1, stream)
            assert rc == 0, lib.rtbhip_last_error()
        return f

    def per_launch(f, reps):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record(); f(); b.record()
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in ev]

    out = {"N": N, "bytes_per_launch": 224 * N}
    bufs = make(N)
    f = launcher(bufs, N)
    for _ in range(5):
        f()
    out["A_steady_200"] = stats(per_launch(f, 200))
    # loop average by one event pair (what bench lines report)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(200):
        f()
    e1.record(); torch.cuda.synchronize()
    out["A_loop_avg_ms"] = e0.elapsed_time(e1) / 200
    # B: fresh buffers per launch
    del bufs
    torch.cuda.empty_cache()
    fresh = []
    for _ in range(12):
        b = make(N)
        torch.cuda.synchronize()
        fresh.append(per_launch(launcher(b, N), 2))
        del b
        torch.cuda.empty_cache()            # give the pages back: the next make() maps new ones
    out["B_fresh_buffers_first_and_second_launch_ms"] = [[round(x, 4) for x in p] for p in fresh]
    # C: clocks and power during a sustained loop
    bufs = make(N)
    f = launcher(bufs, N)
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
                samples.append((time.time(), r.stdout.strip()[:1500]))
            except Exception as e:
                samples.append((time.time(), repr(e)))
    idle = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True).stdout.strip()[:1500]
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    blocks = []
    while time.time() - t0 < 4.0:
        ms = per_launch(f, 100)
        blocks.append({"t": round(time.time() - t0, 2), "min": min(ms), "median": sorted(ms)[50], "max": max(ms)})
    stop.set(); th.join()
    out["C_sustained_blocks_of_100"] = blocks
    out["C_rocm_smi_idle"] = idle
    out["C_rocm_smi_under_load"] = [s for _, s in samples[:6]]
    del bufs
    torch.cuda.empty_cache()
    n2 = 1250000
    bufs = make(n2)
    f = launcher(bufs, n2)
    for _ in range(5):
        f()
    out["D_shard_1.25e6_200"] = stats(per_launch(f, 200))
    print(json.dumps(out))


if __name__ == "__main__":
    main()

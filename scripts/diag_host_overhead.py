#!/usr/bin/env python3
"""Where does host time go in a launch loop?  (visit r2a: 20 back-to-back DHRobot.rne calls on 1.25e6 rows took 2.07 ms each
on the host clock AND on an event pair around the loop, the kernel alone 0.069 ms.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip, ctypes as C
torch.cuda.set_device(0)
N = 1250000
rob = rtbhip.models.DH.Panda()
rng = np.random.default_rng(3)
q, qd, qdd = (torch.from_numpy(rng.normal(size=(N, 7))).cuda() for _ in range(3))
lib = rtbhip.lib()
def loop(fn, k, label):
    fn(); torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for _ in range(k):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    ts = np.array(ts) * 1e6
    print("%-44s total/k %.1f us   host per call: median %.1f  max %.1f us" % (label, tot / k * 1e6, np.median(ts), ts.max()), flush=True)
hold = {}
loop(lambda: hold.__setitem__("t", rob.rne(q, qd, qdd)), 20, "rob.rne, result kept in a dict (bench_extra)")
loop(lambda: rob.rne(q, qd, qdd), 20, "rob.rne, result dropped")
tau = torch.empty((N, 7), dtype=torch.float64, device="cuda")
g = np.array([0, 0, 9.81]); h = rob._dyn_handle(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
loop(lambda: lib.rtbhip_rne(h, p(q), p(qd), p(qdd), N, g.ctypes.data_as(C.c_void_p), None, p(tau), 1, st), 20, "rtbhip_rne through ctypes, preallocated tau")
loop(lambda: torch.empty((N, 7), dtype=torch.float64, device="cuda"), 20, "torch.empty(70 MB) alone")
x = []
loop(lambda: x.append(torch.empty((N, 7), dtype=torch.float64, device="cuda")) or (len(x) > 1 and x.pop(0)), 20, "torch.empty(70 MB), previous one alive")
print(torch.cuda.memory_summary(abbreviated=True)[:1500])

#!/usr/bin/env python3
"""scripts/print_signatures.py [robot ...] -- the structure signatures the dynamics kernels dispatch on, as the host code computes them (through the CPU
replay library, tests/emu: no GPU needed), for the shipped URDF robots (default: all) and the two DH models:

  tree   SegSig  (csrc/tree_device.h, tree.cpp: tree_signature)   per group: class of the constant rotation | translation mask; bit 56 = plain serial chain
         TreeTopo (tree.cpp: tree_topology)                        per group: parent, joint kind, branch slots
  DH     RneSig  (csrc/rne_device.h: rne_signature)               per link: shortcut flags, class of alpha, a == 0, d == 0; no-friction / no-motor bits

Serving another robot with straight-line kernels = one constant in tree_device.h / rne_device.h (what this script prints), one dispatch line in
tree_kernels.hip + tree_dyn_kernels.hip / rne_kernels.hip + dyn_kernels.hip, one in tests/emu (the CPU replay mirrors the launcher), and the robot's name in
tests/test_tree_signature.py / test_rne_signature.py."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import rtbhip
from rtbhip import urdf
from rtbhip._lib import rtbhip_tree_group
import emu_harness as emu
import cpu_backend

CLS = ["G", "I", "RxP", "RxN", "Rx", "RyP", "RyN", "Ry", "RzP", "RzN", "Rz", "pA", "pB"]
L = emu.lib()
L.emu_tree_signature.argtypes, L.emu_tree_signature.restype = [C.POINTER(rtbhip_tree_group), C.c_int32], C.c_uint64
L.emu_tree_signature2.argtypes, L.emu_tree_signature2.restype = [C.POINTER(rtbhip_tree_group), C.c_int32], C.c_uint64
L.emu_tree_topology.argtypes = [C.POINTER(rtbhip_tree_group), C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
L.emu_rne_signature.argtypes, L.emu_rne_signature.restype = [C.c_uint64], C.c_uint64


def table(recs):
    arr = (rtbhip_tree_group * len(recs))()
    for k, r in enumerate(recs):
        arr[k].parent, arr[k].kind, arr[k].flip, arr[k].jindex = r["parent"], r["kind"], r["flip"], r["jindex"]
        arr[k].T[:] = list(np.ascontiguousarray(r["T"]).reshape(16))
        arr[k].m = r["m"]
        arr[k].h[:] = list(r["h"])
        arr[k].I[:] = list(r["I"])
    return arr


names = sys.argv[1:] or sorted(urdf.available())
for n in names:
    recs = urdf.load(n).erobot(()).group_table()
    ng = len(recs)
    arr = table(recs)
    sig, sig2 = L.emu_tree_signature(arr, ng), L.emu_tree_signature2(arr, ng)
    hi, lo = C.c_uint64(), C.c_uint64()
    L.emu_tree_topology(arr, ng, C.byref(hi), C.byref(lo))
    topo = (hi.value << 64) | lo.value
    print("%-11s groups %2d  mass %-5s  SegSig 0x%016x %s%s" % (n, ng, sum(r["m"] for r in recs) > 0, sig, "plain " if (sig >> 56) & 1 else "",
          [(CLS[(sig >> (7 * j)) & 15], (sig >> (7 * j + 4)) & 7) for j in range(min(ng, 8))] if sig else "(more than 16 groups)"))
    if sig2:
        print("%-11s second word 0x%016x %s" % ("", sig2, [(CLS[(sig2 >> (7 * j)) & 15], (sig2 >> (7 * j + 4)) & 7) for j in range(ng - 8)]))
    if topo:
        print("%-11s TreeTopo hi 0x%016x lo 0x%016x  (parent, prismatic, parent slot, save slot) %s" % ("", hi.value, lo.value,
              [(((topo >> (12 * j)) & 15) - 1, (topo >> (12 * j + 4)) & 1, ((topo >> (12 * j + 5)) & 7) - 1, ((topo >> (12 * j + 8)) & 7) - 1) for j in range(ng)]))
with cpu_backend.installed():
    for n in ("Panda", "Puma560"):
        rob = getattr(rtbhip.models.DH, n)()
        h = rob._dyn_handle()
        sig = L.emu_rne_signature(h.value if hasattr(h, "value") else int(h))
        print("DH %-8s links %d  RneSig 0x%016x  no-friction %d  no-motor %d  (flags, alpha class, a == 0, d == 0) %s" % (n, rob.n, sig, (sig >> 62) & 1, (sig >> 61) & 1,
              [((sig >> (7 * j)) & 7, (sig >> (7 * j + 3)) & 3, (sig >> (7 * j + 5)) & 1, (sig >> (7 * j + 6)) & 1) for j in range(rob.n)]))

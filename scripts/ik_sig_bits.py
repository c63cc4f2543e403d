#!/usr/bin/env python3
"""scripts/ik_sig_bits.py -- for every robot k_ik has a signature instantiation for: are the outputs of the specialised kernel bit for bit those of the
general kernel (rtbhip_tune "ik_sig" = 0)?  Prints the number of differing elements per output array and the sustained times."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from rtbhip import urdf
from benchlib import sustained_ms
panda = rtbhip.models.Panda().ets(); panda.qlim = rtbhip.models.PANDA_QLIM
for name, e in (("Panda ETS", panda), ("Panda URDF", urdf.load("Panda").ets()), ("UR5", urdf.load("UR5").ets(end="tool0")), ("UR10", urdf.load("UR10").ets(end="tool0"))):
    lim = np.clip(e.qlim, -2.8, 2.8)
    T = e.eval(torch.from_numpy(np.random.default_rng(1).uniform(lim[0], lim[1], (100000, e.n))).cuda())
    out, ms = {}, {}
    for sig in (1, 0):
        rtbhip.tune("ik_sig", sig)
        f = lambda: e.ik_LM(T, seed=2)
        out[sig] = [x.cpu().numpy() for x in f()]
        ms[sig] = round(sustained_ms(f)[0], 4)
    rtbhip.tune("ik_sig", 1)
    diff = [int((a != b).sum()) for a, b in zip(out[1], out[0])]
    print(json.dumps({"robot": name, "lib": os.path.basename(os.environ.get("RTBHIP_LIB") or "product"), "differing_elements_q_ok_it_se_E": diff,
                      "max_abs_dq": float(np.nanmax(np.abs(out[1][0] - out[0][0]))), "ms_sig": ms[1], "ms_general": ms[0]}), flush=True)

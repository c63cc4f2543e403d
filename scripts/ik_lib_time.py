#!/usr/bin/env python3
"""scripts/ik_lib_time.py -- sustained k_ik timings of ONE library (RTBHIP_LIB selects an A/B build): config 3 (1e5 targets, defaults), the
notebook setting, 1e6 targets, ikine_LM (Python flavour) -- and a checksum of the outputs so that builds can be compared for equality."""
import json, os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np, torch, rtbhip
from benchlib import sustained_ms
ets = rtbhip.models.Panda().ets(); ets.qlim = rtbhip.models.PANDA_QLIM
rng = np.random.default_rng(1)
T5 = ets.eval(torch.from_numpy(rng.uniform(ets.qlim[0], ets.qlim[1], (100000, 7))).cuda())
T6 = ets.eval(torch.from_numpy(rng.uniform(ets.qlim[0], ets.qlim[1], (1000000, 7))).cuda())
tunes = os.environ.get("RTBHIP_TUNE", "")          # "key=value,key=value": rtbhip_tune settings of this run (an A/B inside one library)
for kv in filter(None, tunes.split(",")):
    k, v = kv.split("=")
    rtbhip.tune(k, int(v))
out = {"lib": (os.environ.get("RTBHIP_LIB") or "product") + ((" " + tunes) if tunes else "")}
for name, T, kw in (("config3", T5, {}), ("notebook", T5, {"k": 0.1, "joint_limits": False}), ("1e6", T6, {})):
    res = {}
    def run():
        res["o"] = ets.ik_LM(T, seed=2, **kw)
    run()
    ms, reps, warm = sustained_ms(run)
    q, ok, it, se, E = res["o"]
    h = hashlib.sha256(b"".join(x.cpu().numpy().tobytes() for x in (ok, it, se))).hexdigest()[:12]
    hq = hashlib.sha256(q.cpu().numpy().tobytes() + E.cpu().numpy().tobytes()).hexdigest()[:12]          # every bit of q and E
    out[name] = {"ms": round(ms, 4), "ok": float(ok.float().mean()), "its": int(it.sum()), "counts_sha": h, "bits_sha": hq, "q_sum": float(torch.nan_to_num(q).sum())}
print(json.dumps(out), flush=True)

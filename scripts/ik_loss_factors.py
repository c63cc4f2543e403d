#!/usr/bin/env python3
"""scripts/ik_loss_factors.py [--launches K] -- config 3's IK call K times (+ one launch with the kernel's diagnostic counters), one JSON line.
Run plainly it prints the loss factors benchsecondary.py puts on the bench line; run under
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d DIR -o sq -- python scripts/ik_loss_factors.py --launches 6
the counter file gives SQ_INSTS_VALU per k_ik dispatch, and   python scripts/ik_loss_factors.py --digest DIR OUT.json   divides it by the wave
iterations this script printed (its stdout saved as DIR/run.json) -> profiles/r06_ik_sq.json, the committed constant of the bench line."""
import argparse, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=6)
ap.add_argument("--digest", nargs=2)
args = ap.parse_args()
if args.digest:
    d, outp = args.digest
    run = json.load(open(os.path.join(d, "run.json")))
    vals, waves = [], []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ik<" in r["Kernel_Name"]:
                (vals if r["Counter_Name"] == "SQ_INSTS_VALU" else waves).append(float(r["Counter_Value"]))
    assert vals, "no k_ik dispatch in the counter file"
    # SQ_INSTS_VALU counts per wave-instruction; one k_ik dispatch of the flat schedule does the whole call's iterations
    per_launch = sum(vals) / len(vals)
    out = {"visit": os.path.basename(os.path.normpath(d)), "sq_insts_valu_per_launch": per_launch, "dispatches_counted": len(vals),
           "sq_waves_per_launch": (sum(waves) / len(waves)) if waves else None, "wave_iterations_per_launch": run["wave_iterations"],
           "valu_per_wave_iteration": per_launch / run["wave_iterations"], "kernel": run["kernel"],
           "note": "scheduling passes, the prologue and the final emit are amortised into the figure"}
    json.dump(out, open(outp, "w"), indent=1)
    print(json.dumps(out))
    sys.exit(0)
import numpy as np
import torch
import rtbhip
import benchsecondary as bs
from benchlib import sustained_ms
ets = rtbhip.models.Panda().ets()
ets.qlim = rtbhip.models.PANDA_QLIM
qs = torch.from_numpy(np.random.default_rng(1).uniform(ets.qlim[0], ets.qlim[1], (100000, 7))).cuda()
Tep = ets.eval(qs)
res = {}


def run():
    res["out"] = ets.ik_LM(Tep, seed=2)


for _ in range(args.launches):
    run()
torch.cuda.synchronize()
ms, _, _ = sustained_ms(run)
its = float(res["out"][2].sum())
out = bs.ik_loss_factors(run, its, ms)
out.update({"kernel_avg_ms": ms, "lm_iterations": its, "kernel": "k_ik<7,0,13,kIkSigPandaETS> (flat schedule)"})
print(json.dumps(out))

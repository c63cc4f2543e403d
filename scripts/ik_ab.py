#!/usr/bin/env python3
"""scripts/ik_ab.py -- k_ik at BASELINE configs[2] (1e5 Panda targets, ik_LM defaults) under the scheduler's A/B knobs (rtbhip_tune), measured in
the STEADY STATE (benchlib.sustained_ms: the earlier tuning of these knobs used 3-launch averages, i.e. the boost clock; under sustained load the
part is power-limited and discarded speculation costs clock).  Variants interleaved over `rounds` rounds; one JSON line per variant."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
import numpy as np
import torch
import rtbhip
from benchlib import sustained_ms

# "shipped" = the ROUND-3 defaults (flat 4/8, fresh share 50 %): the baseline of these sweeps.  Round 4 ships fresh 100 %, first chunk automatic (8 / 4).
DEFAULTS = {"ik_flat": 1, "ik_flat_l0": 4, "ik_flat_len": 8, "ik_waves_per_cu": 8, "ik_pass_mask": 3, "ik_fresh_pct": 50, "ik_spec_policy": 0, "ik_share": 0}
ROUND4 = {"ik_flat_l0": 0, "ik_flat_len": 0, "ik_fresh_pct": 100, "ik_unit_we": 1, "ik_plain": 1}
VARIANTS = [("shipped", {}), ("plain", {"ik_flat": 0}), ("flat 4/16", {"ik_flat_len": 16}), ("flat 8/16", {"ik_flat_l0": 8, "ik_flat_len": 16}),
            ("flat 2/8", {"ik_flat_l0": 2}), ("flat 4/4", {"ik_flat_len": 4}), ("flat 6/8", {"ik_flat_l0": 6}), ("flat 3/6", {"ik_flat_l0": 3, "ik_flat_len": 6}),
            ("flat 4/8, 4 waves/CU", {"ik_waves_per_cu": 4}), ("flat 4/8, 6 waves/CU", {"ik_waves_per_cu": 6}), ("plain, 4 waves/CU", {"ik_flat": 0, "ik_waves_per_cu": 4}),
            ("pass every 2", {"ik_pass_mask": 1}), ("pass every 8", {"ik_pass_mask": 7}), ("fresh 25 %", {"ik_fresh_pct": 25}), ("fresh 100 %", {"ik_fresh_pct": 100}),
            ("failure-weighted speculation", {"ik_spec_policy": 1}), ("plain + sharing", {"ik_flat": 0, "ik_share": 1})]
if os.environ.get("IK_AB_SET") == "2":          # second sweep: around fresh 100 %
    VARIANTS = [("shipped", {})] + [("fresh %d, flat %d/%d" % (f, a, b), {"ik_fresh_pct": f, "ik_flat_l0": a, "ik_flat_len": b})
                                    for f in (75, 100, 140) for a, b in ((4, 8), (6, 8), (5, 8), (6, 10), (6, 12), (8, 8), (8, 12), (5, 6))] + \
               [("fresh 100, plain", {"ik_fresh_pct": 100, "ik_flat": 0}), ("fresh 100, plain + sharing", {"ik_fresh_pct": 100, "ik_flat": 0, "ik_share": 1}),
                ("fresh 100, flat 6/8, pass every 8", {"ik_fresh_pct": 100, "ik_flat_l0": 6, "ik_pass_mask": 7}),
                ("fresh 100, flat 6/8, 10 waves/CU", {"ik_fresh_pct": 100, "ik_flat_l0": 6, "ik_waves_per_cu": 10})]
if os.environ.get("IK_AB_SET") == "4":          # third sweep: around fresh 100 %, first chunk 8
    VARIANTS = [("shipped", {})] + [("fresh %d, flat %d/%d" % (f, a, b), {"ik_fresh_pct": f, "ik_flat_l0": a, "ik_flat_len": b})
                                    for f in (90, 100, 110, 120) for a, b in ((8, 8), (8, 12), (7, 8), (10, 8), (8, 6), (10, 10), (12, 8), (9, 9))]
if os.environ.get("IK_AB_SET") == "5":
    VARIANTS = [("shipped", {}), ("fresh 100, flat 6/8", {"ik_fresh_pct": 100, "ik_flat_l0": 6}), ("fresh 100, flat 8/8", {"ik_fresh_pct": 100, "ik_flat_l0": 8}),
                ("fresh 100, flat 8/12", {"ik_fresh_pct": 100, "ik_flat_l0": 8, "ik_flat_len": 12}), ("fresh 100, flat 10/8", {"ik_fresh_pct": 100, "ik_flat_l0": 10})]
if os.environ.get("IK_AB_SET") == "6":          # what round 4 ships against what round 3 shipped
    VARIANTS = [("round-3 knobs", {}), ("round-4 knobs (fresh 100 %, first chunk automatic)", dict(ROUND4))]
if os.environ.get("IK_AB_SET") == "7":          # unit-weight kernel instantiations against the weighted ones (same library)
    DEFAULTS = dict(DEFAULTS, **ROUND4)
    VARIANTS = [("weighted kernel (ik_unit_we = 0)", {"ik_unit_we": 0}), ("unit-weight kernel, general walk", {"ik_unit_we": 1, "ik_plain": 0}),
                ("unit-weight kernel, plain-revolute walk", {"ik_unit_we": 1, "ik_plain": 1})]
if os.environ.get("IK_AB_SET") == "8":          # later-chunk length re-measured on the final iteration (round 4, after the kernel work)
    DEFAULTS = dict(DEFAULTS, **ROUND4)
    VARIANTS = [("flat auto/auto (shipped)", {}), ("flat auto/8", {"ik_flat_len": 8}), ("flat auto/10", {"ik_flat_len": 10}), ("flat auto/12", {"ik_flat_len": 12}), ("flat auto/16", {"ik_flat_len": 16}),
                ("flat 10/10", {"ik_flat_l0": 10, "ik_flat_len": 10}), ("flat 8/12, fresh 90", {"ik_flat_l0": 8, "ik_flat_len": 12, "ik_fresh_pct": 90})]
if os.environ.get("IK_AB_SET") == "9":          # the other scheduler knobs once more, on the final kernel and the automatic chunk lengths
    DEFAULTS = dict(DEFAULTS, **ROUND4)
    VARIANTS = [("shipped", {}), ("pass every 2", {"ik_pass_mask": 1}), ("pass every 8", {"ik_pass_mask": 7}), ("fresh 90", {"ik_fresh_pct": 90}),
                ("fresh 110", {"ik_fresh_pct": 110}), ("6 waves/CU", {"ik_waves_per_cu": 6}), ("10 waves/CU", {"ik_waves_per_cu": 10}),
                ("first chunk 10", {"ik_flat_l0": 10}), ("first chunk 6", {"ik_flat_l0": 6})]
if os.environ.get("IK_AB_SET") == "3":          # other batch sizes / settings: does the candidate hold?
    VARIANTS = [("shipped", {}), ("fresh 100", {"ik_fresh_pct": 100}), ("fresh 100, flat 6/8", {"ik_fresh_pct": 100, "ik_flat_l0": 6}), ("fresh 140, flat 6/8", {"ik_fresh_pct": 140, "ik_flat_l0": 6})]

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ets = rtbhip.models.Panda().ets()
    ets.qlim = rtbhip.models.PANDA_QLIM
    qs = torch.from_numpy(np.random.default_rng(1).uniform(ets.qlim[0], ets.qlim[1], (N, 7))).cuda()
    Tep = ets.eval(qs)
    res = {}

    kw = {"k": 0.1, "joint_limits": False} if os.environ.get("IK_AB_NOTEBOOK") == "1" else {}

    def run():
        res["o"] = ets.ik_LM(Tep, seed=2, **kw)
    base = None
    out = {name: [] for name, _ in VARIANTS}
    for r in range(rounds):
        for name, kv in VARIANTS:
            for k, v in dict(DEFAULTS, **kv).items():
                rtbhip.tune(k, v)
            run()
            ms, reps, warm = sustained_ms(run)
            o = [x.cpu().numpy() for x in res["o"]]
            if base is None:
                base = o
            same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(base, o))
            out[name].append((ms, same))
    for k, v in dict(DEFAULTS, **ROUND4).items():
        rtbhip.tune(k, v)
    for name, kv in VARIANTS:
        print(json.dumps({"variant": name, "tune": kv, "n": N, "sustained_ms": [round(m, 4) for m, _ in out[name]], "outputs_equal_shipped": all(s for _, s in out[name])}), flush=True)


if __name__ == "__main__":
    main()

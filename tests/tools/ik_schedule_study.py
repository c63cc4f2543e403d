"""tests/tools/ik_schedule_study.py -- TEST INFRASTRUCTURE / analysis, not part of the product.

Two CPU experiments on k_ik's speculative scheduler at BASELINE config 3 (1e5 Panda targets, ik_LM defaults, 2048 single-wave workgroups):

  replay   the kernel's own scheduler code (tests/emu replays ik_device.h / ik_kernels.hip wave by wave) -- wave iterations max / mean,
           scheduling passes, running lane-iterations, for the plain and the flat schedule.  Reproduces the per-wave counters the GPU run
           wrote (profiles/r03_c_ik_occupancy.txt) to a few percent, so schedule changes can be costed here before they go to the GPU.
  floor    an IDEALISED scheduler on the same work: all lanes one pool (no wave boundaries, no cost for a pass), every free lane given
           the not-yet-started search most likely to be needed (posterior from the batch's own success statistics), speculative
           searches cancelled at the first pass after an earlier one succeeded.  What it needs is a lower bound for any schedule that
           cannot know in advance how many restarts a target takes.

  python tests/tools/ik_schedule_study.py replay [N waves]      python tests/tools/ik_schedule_study.py floor [lanes pass_period]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import rtbhip                     # noqa: E402  (host-side classes only: chain description, joint limits)
import emu_harness as emu         # noqa: E402


def targets(N):
    """bench_extra.py's IK workload: poses of uniform draws inside the Panda's joint limits (seed 1), solver seed 2."""
    ets = rtbhip.models.Panda().ets()
    ets.qlim = rtbhip.models.PANDA_QLIM
    lim = ets._limits(False)
    qs = np.random.default_rng(1).uniform(lim[0], lim[1], (N, 7))
    return ets, emu.kin(ets, qs, want=("T",))[0]


def replay(N, waves):
    ets, Tep = targets(N)
    useful = None
    for name, env in (("plain", {}), ("flat 4/8 (shipped)", {"EMU_IK_FLAT": "1"}), ("flat 4/16", {"EMU_IK_FLAT": "1", "EMU_IK_FLAT_LEN": "16"}),
                      ("flat 8/16", {"EMU_IK_FLAT": "1", "EMU_IK_FLAT_L0": "8", "EMU_IK_FLAT_LEN": "16"}),
                      ("flat 12/24", {"EMU_IK_FLAT": "1", "EMU_IK_FLAT_L0": "12", "EMU_IK_FLAT_LEN": "24"})):
        for k in ("EMU_IK_FLAT", "EMU_IK_FLAT_L0", "EMU_IK_FLAT_LEN"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["EMU_IK_PASS_MASK"] = "3"            # the kernel's pass period (every 4th iteration)
        st = [0, 0, 0, 0]
        t0 = time.time()
        q, ok, it, se, E = emu.ik(ets, Tep, seed=2, waves=waves, stats=st)
        useful = int(it.sum())
        slots = st[1] * 64
        print("%-20s wave iterations max %3d mean %6.1f   passes/wave %5.1f   lane slots %.3e = useful %4.1f %% + discarded %4.1f %% + idle %4.1f %%   (%.0f s)"
              % (name, st[0], st[1] / waves, st[3] / waves, slots, 100.0 * useful / slots, 100.0 * (st[2] - useful) / slots, 100.0 * (slots - st[2]) / slots,
                 time.time() - t0))
    print("useful lane-iterations %d = %.1f iterations of %d lanes" % (useful, useful / (64.0 * waves), 64 * waves))
    for k in ("EMU_IK_FLAT", "EMU_IK_FLAT_L0", "EMU_IK_FLAT_LEN", "EMU_IK_PASS_MASK"):
        os.environ.pop(k, None)
    return useful


def floor(N, lanes, period):
    """Returns (iterations to the last result, useful lane-iterations, discarded lane-iterations)."""
    ets, Tep = targets(N)
    q, ok, it, se, E = emu.ik(ets, Tep, seed=2)          # the sequential specification: searches and iterations each target needs
    need = np.where(ok == 1, se, 100).astype(np.int64)
    succ_at = np.where(ok == 1, se, 10 ** 9)
    slen = np.maximum(1, np.rint(it / need)).astype(np.int64)       # model: a target's searches are equally long
    cnt = np.bincount(np.where(ok == 1, se, 101), minlength=102)[1:102].astype(float)
    surv = cnt[::-1].cumsum()[::-1]
    haz = np.where(surv[:100] > 0, cnt[:100] / np.maximum(surv[:100], 1), 0.0)          # P(search j+1 succeeds | j failed)
    clq = np.concatenate([[0.0], np.log(np.maximum(1e-9, 1.0 - haz)).cumsum()])
    nxt = np.zeros(N, np.int64); infl = np.zeros(N, np.int64); fin_before = np.zeros(N, np.int64)
    succeeded = np.zeros(N, bool); resolved = np.zeros(N, bool)
    lane_t = -np.ones(lanes, np.int64); lane_k = np.zeros(lanes, np.int64); lane_end = np.zeros(lanes, np.int64)
    tau = useful = discarded = 0
    while not resolved.all():
        idx = np.nonzero((lane_t >= 0) & (lane_end <= tau))[0]
        if len(idx):
            t, k = lane_t[idx], lane_k[idx]
            np.subtract.at(infl, t, 1)
            succeeded[t[(k + 1) == succ_at[t]]] = True
            needed = (k + 1) <= np.minimum(succ_at[t], 100)
            np.add.at(fin_before, t[needed], 1)
            useful += int(slen[t[needed]].sum()); discarded += int(slen[t[~needed]].sum())
            lane_t[idx] = -1
        bi = np.nonzero(lane_t >= 0)[0]
        if len(bi):                                   # cancel what an earlier success has made pointless
            t, k = lane_t[bi], lane_k[bi]
            ci = bi[succeeded[t] & ((k + 1) > succ_at[t])]
            if len(ci):
                discarded += int((slen[lane_t[ci]] - (lane_end[ci] - tau)).sum())
                np.subtract.at(infl, lane_t[ci], 1)
                lane_t[ci] = -1
        resolved = fin_before >= np.minimum(succ_at, 100)
        free = np.nonzero(lane_t < 0)[0]
        pos = 0
        while pos < len(free):
            c = np.nonzero((~succeeded) & (nxt < 100) & (~resolved))[0]
            if len(c) == 0:
                break
            lp = clq[nxt[c]] - clq[nxt[c] - infl[c]]          # log P(every search of the target still in flight fails)
            sel = c[np.argsort(-lp, kind="stable")][:len(free) - pos]
            ln = free[pos:pos + len(sel)]
            lane_t[ln] = sel; lane_k[ln] = nxt[sel]; lane_end[ln] = tau + slen[sel]
            nxt[sel] += 1; infl[sel] += 1
            pos += len(sel)
        tau += period
    print("idealised pool of %d lanes, a pass every %d iterations: %d iterations to the last result; useful %.3e + discarded %.3e lane-iterations;"
          " useful alone = %.1f iterations" % (lanes, period, tau, useful, discarded, it.sum() / lanes))
    return tau, useful, discarded


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "replay"
    if what == "replay":
        replay(int(sys.argv[2]) if len(sys.argv) > 2 else 100000, int(sys.argv[3]) if len(sys.argv) > 3 else 2048)
    else:
        lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
        for period in ([int(sys.argv[3])] if len(sys.argv) > 3 else [4, 2, 1]):
            floor(100000, lanes, period)

#!/usr/bin/env python3
"""bench_extra.py -- secondary measurements of the other rows of SURVEY.md 8 (NOT the headline; the
driver only runs bench.py).  One JSON object per line:
   rne    BASELINE configs[3] per-GPU share: DH Panda inverse dynamics, N (q,qd,qdd) triples
   ik     BASELINE configs[2]: ik_LM over 1e5 reachable targets, defaults (chan, k=1)
   fleet  config-5-shaped mixed fleet through one launch
Each leg times K launches with HIP events on the launch stream and, for rne/ik, the reference's own
CPU path (oracle/_ref) on a bounded sample."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]


def ev_time(fn, steps, warmup):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="rne,ik,fleet")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n-rne", type=int, default=1250000)
    ap.add_argument("--n-ik", type=int, default=100000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--tune", action="append", default=[])
    args = ap.parse_args()
    import numpy as np
    import torch
    import rtbhip
    from oracle import chains
    torch.cuda.set_device(0)
    for kv in args.tune:
        k, v = kv.split("=")
        rtbhip.tune(k, int(v))
    what = args.what.split(",")

    if "rne" in what:
        N = args.n_rne
        rob = rtbhip.models.DH.Panda()
        tab = chains.panda_dh()
        rng = np.random.default_rng(3)
        qh = rng.uniform(tab.qlim[:, 0], tab.qlim[:, 1], (N, 7))
        qdh, qddh = rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
        q, qd, qdd = (torch.from_numpy(x).cuda() for x in (qh, qdh, qddh))
        avg, best = ev_time(lambda: rob.rne(q, qd, qdd), args.steps, 3)
        line = {"metric": "triples/sec (DH Panda rne)", "value": N / (avg * 1e-3), "unit": "triples/s", "n": N,
                "kernel_avg_ms": avg, "kernel_min_ms": best,
                "roofline": {"bound": "hbm", "achieved": 224.0 * N / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": 224.0 * N / (avg * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": 224 * N}}
        if not args.no_cpu:
            from oracle import ref_harness
            if ref_harness.available():
                ref = ref_harness.RefRNE(tab.L24(), 1)
                n = 100000
                t0 = time.perf_counter(); tau = ref.rne(qh[:n], qdh[:n], qddh[:n]); dt = time.perf_counter() - t0
                g = rob.rne(q[:n], qd[:n], qdd[:n]).cpu().numpy()
                line["cpu_baseline"] = {"value": n / dt, "unit": "triples/s", "cores": 1, "kind": "reference",
                                        "sample": "frne.frne per-row loop (DHRobot.rne) over the first %d triples" % n,
                                        "max_rel_err_gpu_vs_cpu": float(np.abs(g - tau).max() / np.abs(tau).max())}
        print(json.dumps(line), flush=True)

    if "ik" in what:
        N = args.n_ik
        ets = rtbhip.models.Panda().ets()
        ets.qlim = chains.PANDA_QLIM
        ch = chains.panda_ets(with_limits=True)
        rng = np.random.default_rng(1)
        qs = torch.from_numpy(rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7))).cuda()
        Tep = ets.eval(qs)
        res = {}
        def run():
            res["out"] = ets.ik_LM(Tep, seed=2)
        avg, best = ev_time(run, max(3, args.steps // 4), 1)
        q, ok, it, se, E = res["out"]
        line = {"metric": "solves/sec (Panda ik_LM chan k=1, joint limits, ilimit 30 slimit 100 tol 1e-6)",
                "value": N / (avg * 1e-3), "unit": "solves/s", "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best,
                "success_rate": float(ok.float().mean()), "mean_iterations": float(it.float().mean()),
                "max_iterations": int(it.max()), "lm_iterations_per_s": float(it.sum()) / (avg * 1e-3)}
        if not args.no_cpu:
            from oracle import ref_harness
            if ref_harness.available():
                ref = ref_harness.RefETS(ch)
                n = 2000
                Th = Tep[:n].cpu().numpy()
                t0 = time.perf_counter()
                out = [ref.ik_LM(Th[i]) for i in range(n)]
                dt = time.perf_counter() - t0
                line["cpu_baseline"] = {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "reference",
                                        "sample": "IK_LM_c loop over the first %d targets" % n,
                                        "success_rate": float(np.mean([o[1] for o in out])),
                                        "mean_iterations": float(np.mean([o[2] for o in out]))}
        print(json.dumps(line), flush=True)

    if "fleet" in what:
        rng = np.random.default_rng(4)
        panda = rtbhip.models.Panda().ets()
        puma = rtbhip.models.DH.Puma560().ets()
        pdh = rtbhip.models.DH.Panda().ets()
        chs = [panda, puma, pdh, panda, puma, pdh, panda, puma]
        N = 250000
        qs = [torch.from_numpy(rng.uniform(-3, 3, (N, c.n))).cuda() for c in chs]
        avg, best = ev_time(lambda: rtbhip.fleet_fkine_jacob(chs, qs), args.steps, 2)
        byts = sum(N * (8 * c.n + 128 + 48 * c.n) for c in chs)
        line = {"metric": "configurations/sec (mixed fleet of %d chains, one launch)" % len(chs),
                "value": N * len(chs) / (avg * 1e-3), "unit": "configurations/s", "kernel_avg_ms": avg,
                "roofline": {"bound": "hbm", "achieved": byts / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                             "frac": byts / (avg * 1e-3) / 1e9 / 8000.0}}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

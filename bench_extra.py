#!/usr/bin/env python3
"""bench_extra.py -- secondary measurements of the other rows of SURVEY.md 8 (NOT the headline; the
driver only runs bench.py).  One JSON object per line:
   rne    BASELINE configs[3] per-GPU share: DH Panda inverse dynamics, N (q,qd,qdd) triples
   ik     BASELINE configs[2]: ik_LM over 1e5 reachable targets, defaults (chan, k=1)
   fleet  config-5-shaped mixed fleet through one launch
Each leg times K launches with HIP events on the launch stream and, for rne/ik, the reference's own
CPU path (oracle/_ref) on a bounded sample.

Multi-GPU (`--gpus N`, self-spawning like bench.py; legs rne, ik and fleet only):
   rne    BASELINE configs[3] as stated: N = 1e7 triples in total, split into contiguous row blocks by
          rtbhip_shard_range, one block per rank ("scaling": "strong"); the gather of tau (all_gather over
          RCCL/xGMI) is timed separately as gather_ms
   ik     BASELINE configs[2]: the 1e5 targets in row blocks over the ranks
   fleet  BASELINE configs[4]: 16 URDF arms x 1e6 configurations, every arm's batch split by rows over the ranks"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]

from benchlib import protect_stdout, HBM_PEAK_GBS, FP64_VALU_PEAK_TFLOPS, Ranks, spawn_ranks_if_needed, bench_argv, ensure_library, pmc_traffic, sustained_counts  # noqa: E402


def ev_time(fn, steps, warmup):
    """(average, minimum) device-side duration of `fn` in the STEADY STATE: >= 30 ms of warm-up launches, then per-launch HIP-event pairs over
    >= 30 ms of launches (at most 200), no synchronise in between (benchlib.sustained_counts says why: boost clock after an idle gap, then a
    throttling transient, profiles/r04_rne_1e7.txt)."""
    import torch
    for _ in range(warmup):
        fn()
    warm, reps = sustained_counts(fn)
    steps = max(steps, min(reps, 200))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    import gc
    gc.collect()                      # no collector pause (40-60 ms here) between a launch and its closing event
    was = gc.isenabled()
    gc.disable()
    for _ in range(warm):
        fn()
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    if was:
        gc.enable()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return sum(ms) / len(ms), ms[0]


# fp64 operations of ONE Levenberg-Marquardt iteration of the 7-joint Panda as k_ik executes it (fused multiply-add = 2):
# FK + Jacobian walk incl. 7 sincos ~0.60 k, angle-axis error + E ~0.12 k, J^T W J + g (lower triangle) ~0.46 k,
# 7x7 LDL^T factor + solves ~0.25 k, update / wrap / limit test ~0.05 k  (the reference's formulation needs ~5.5 k)
IK_FLOPS_PER_ITERATION = 1480.0


def ik_roofline(lm_iterations_per_s):
    """IK is not HBM-bound (204 B of I/O per target against ~75 iterations): the roof is the fp64 vector rate.  `achieved`
    counts only the iterations the REFERENCE's sequential loops would have run (reported iteration counts), i.e. discarded
    speculative searches and idle lanes count against the kernel."""
    tf = lm_iterations_per_s * IK_FLOPS_PER_ITERATION / 1e12
    return {"bound": "fp64-valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
            "flops_per_iteration": IK_FLOPS_PER_ITERATION, "kernel": "k_ik<7,0>"}


# ALGORITHMIC fp64 operations per unit of the issue-bound secondary kernels (fused multiply-add = 2), derived in DESIGN.md section 5 from the
# recursion's own budget -- the best formulation known here (the one the kernels implement), NOT what a kernel happens to execute:
#   full Newton-Euler link-pass   135 fp64 operations-as-instructions = 270 flop
#   acceleration-only link-pass   (all angular velocities zero: no w x (w x r) terms)  forward R^T wd 8, + z qdd 1, wd x p* 6, + vd 3, R^T(..) 8,
#                                 vd_c 6, F 3, N = I wd 9; backward R f 8, f 3, N + r_c x F 6, n 17, projection 4 = 82 = 164 flop
#   sincos                        30 operations = 60 flop per joint (Cody-Waite reduction + two degree-5 polynomials + quadrant selects)
#     gravload    7 acceleration-only link-passes at rest (gravity as the base's acceleration) + 7 sincos       1 148 +   420 =  1 568
#     inertia     column i = one acceleration-only pass over links i..n, mirrored: n (n + 1) / 2 = 28 link-passes                4 592 +   420 =  5 012
#   two-field link-pass           (Dynamics.coriolis, column k = B(qd, e_k): both velocity fields through one recursion, every product of two
#                                 velocities taken both ways round -- csrc/dyn_device.h rne_bilinear_core)  forward R^T w_u 8, R^T w_w 8, w' 2,
#                                 wd' 12, a' 38 (wd x p*, two w x (w x p*), R^T), v_c 30, F 3, N 39 (three I w, two crosses); backward 38 = 178 = 356 flop
#   prefix link-step              (links before the column's own: only w_u advances, 9, and the force is handed down, 22) = 31 = 62 flop
#     coriolis    column k = one two-field pass over links k..n after the prefix: 28 two-field link-passes + 21 prefix steps      9 968 + 1 302 + 420 = 11 690
#                 (rounds 1-3 ran 2 full passes per column -- the polar form, priced 26 880; the reference runs 28 full passes)
#     accel       1 full pass + the inertia columns + LDL^T solve (n^3 / 3 + 2 n^2 = 212)                         1 890 + 4 592 + 212 + 420 =  7 114
#     tree_*      the same link-pass budgets on the 6 link groups of the UR5 (the spatial-vector recursion of Robot.rne is priced at the DH
#                 budget: a lower bound of its arithmetic): rne 1 620 + 360 = 1 980, inertia 21 x 164 + 360 = 3 804, coriolis 21 x 356 + 15 x 62 + 360 = 8 766,
#                 accel 1 620 + 3 444 + 144 + 360 = 5 568
#   FK + Jacobian walk of the 7-joint chain = 600 flop, then
#   (rounds 1-5 priced the pair loops the definitions suggest: jacob0_dot 1 755, manipulability 1 050 with a row-exchanging LU, jacobm 3 000 with 49
#   Hessian-block contractions.  Round 6 found the O(n) running-sum forms -- csrc/diff_device.h -- and the budgets are those of the better formulation now;
#   `frac_at_round5_pricing` on the line keeps the old yardstick beside the new one: a faster kernel must not look slower because its algorithm improved.)
#     jacob0_dot      + prefix / suffix sums (6 FMAs per joint) + three cross products and three adds per joint (30): 7 x 42 = 294       ->   894
#     manipulability  + J J^T (21 entries x 7 FMAs = 294) + LDL^T pivots of the 6 x 6 Gram matrix (116) + sqrt                          -> 1 010
#     jacobm          + J J^T 294 + LDL^T 100 + 7 back-substitutions (462) + per joint three cross products, sums and two dot products (45 x 7) + m   -> 1 780
#   ROUND 5 -- the DH Panda re-priced.  Its link table is structured (alpha = 0 / +-pi/2, centres of mass at the link origins, a or d zero on most links, no
#   friction, no motor inertia) and the kernels now exploit that at compile time (rne_device.h: RneSig), so the budgets above -- those of a GENERAL link --
#   are no longer the best formulation known for THIS robot: a fraction quoted against them would count work nobody has to do (accel: 0.95).  The
#   structured link-passes, from the signature kernels' own SQ counts less trig and staging (profiles/r05_t_sq_digest.txt): full 92 operations = 184 flop
#   (general 135), acceleration-only 66 = 132 flop (82), two-field passes 0.76 of the general ones:
#     gravload 7 x 132 + 420 = 1 344; inertia 28 x 132 + 420 = 4 116; coriolis 0.76 x (9 968 + 1 302) + 420 = 8 985; accel 7 x 184 + 28 x 132 + 212 + 420 = 5 616
#   (the UR5 tree stays priced at the general DH budget: its signature kernels still execute more than that, 166 operations per group-pass)
ROUND5_FLOPS_PER_UNIT = {"jacob0_dot": 1755, "manipulability": 1050, "jacobm": 3000}
ALGO_FLOPS_PER_UNIT = {"gravload": 1344, "inertia": 4116, "coriolis": 8985, "accel": 5616, "tree_ur5": 1980, "jacob0_dot": 894,
                       "manipulability": 1010, "jacobm": 1780, "tree_inertia_ur5": 3804, "tree_coriolis_ur5": 8766, "tree_accel_ur5": 5568,
                       "tree_gravload_ur5": 1344}        # 6 acceleration-only link-passes + 6 sincos
# VALU instructions per lane actually EXECUTED (SQ_INSTS_VALU / SQ_WAVES; the dynamics kernels' structure-signature instantiations of round 5:
# profiles/r05_t_sq_digest.txt -- the general kernels execute 833 / 3433 / 5954 / 3377 and 1929 / 4345 / 10864 / 7216 / 1275, profiles/r04_u_sq_tree_dyn.txt,
# r04_v_sq_dyn.txt; the kinematics consumers: profiles/r03_a_sq_summary.txt): reported beside
# the roofline as `valu_issue_util` (share of the chip's fp64 issue slots the kernel fills) -- a diagnostic, not a roofline: a kernel that
# executed more instructions for the same answer would score higher on it.
# (jacob0_dot / manipulability / jacobm: round 6's running-sum forms, profiles/r06_z_sq_digest.txt -- rounds 3-5: 1692 / 1512 / 2732)
VALU_PER_UNIT = {"gravload": 691, "inertia": 2958, "coriolis": 4682, "accel": 2839, "tree_ur5": 1055, "jacob0_dot": 1361,
                 "manipulability": 1182, "jacobm": 1897,
                 "tree_inertia_ur5": 2638, "tree_coriolis_ur5": 5231, "tree_accel_ur5": 3246, "tree_gravload_ur5": 706}      # profiles/r06_z_sq_digest.txt (round 5: r05_t_sq_digest.txt)


def valu_roofline(key, units_per_s, kernel, hbm_bytes_per_unit):
    """The roof that BINDS the kernel: algorithmic bytes / 8 TB/s against algorithmic flops / 78.6 TFLOP/s per unit, whichever takes longer (the other
    fraction is carried beside it).  jacob0_dot (448 B against 1 755 flop per configuration) is HBM-bound by that rule; rounds 1-3 priced it as fp64."""
    flops = ALGO_FLOPS_PER_UNIT[key]
    tf = flops * units_per_s / 1e12
    gbs = hbm_bytes_per_unit * units_per_s / 1e9
    src = ("DESIGN.md section 5 (general link: full / acceleration-only / two-field link-passes at 270 / 164 / 356 flop; the DH Panda's structured table: 184 / 132 / "
           "0.76 x; 60 flop per sincos; FK + Jacobian 600 flop + the consumer's own products)")
    if hbm_bytes_per_unit / (HBM_PEAK_GBS * 1e9) >= flops / (FP64_VALU_PEAK_TFLOPS * 1e12):
        out = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "algorithmic_bytes_per_unit": hbm_bytes_per_unit, "algorithmic_flops_per_unit": flops, "flops_source": src,
               "kernel": kernel, "hbm_GBs": gbs, "fp64_valu_frac": tf / FP64_VALU_PEAK_TFLOPS}
    else:
        out = {"bound": "fp64-valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
               "algorithmic_flops_per_unit": flops, "flops_source": src,
               "kernel": kernel, "hbm_GBs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS}
    if key in ROUND5_FLOPS_PER_UNIT:
        out["frac_at_round5_pricing"] = ROUND5_FLOPS_PER_UNIT[key] * units_per_s / 1e12 / FP64_VALU_PEAK_TFLOPS
    instr = VALU_PER_UNIT.get(key)
    if instr is not None:
        out["valu_issue_util"] = 2.0 * instr * units_per_s / 1e12 / FP64_VALU_PEAK_TFLOPS
        out["valu_instructions_per_unit_measured"] = instr
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="rne,ik,fleet,dyn,tree,kin,poe,graph,servo")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-rne", type=int, default=0, help="TOTAL triples over all ranks (default: 1.25e6 on one GPU = the "
                    "per-GPU share of config 4; 1e7 = config 4 itself with --gpus > 1)")
    ap.add_argument("--n-ik", type=int, default=100000)
    ap.add_argument("--n-fleet", type=int, default=1000000)
    ap.add_argument("--n-dyn", type=int, default=1000000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--tune", action="append", default=[])
    argv = bench_argv()
    args = ap.parse_args(argv)
    spawn_ranks_if_needed(args.gpus, os.path.abspath(__file__), argv)
    protect_stdout()
    import numpy as np
    import torch
    ensure_library(ROOT)
    import rtbhip
    rk = Ranks()
    rank, world = rk.rank, rk.world
    for kv in args.tune:
        k, v = kv.split("=")
        rtbhip.tune(k, int(v))
    what = args.what.split(",")
    if world > 1:
        what = [w for w in what if w in ("rne", "ik", "fleet")]    # the legs BASELINE shards; the rest are single-GPU figures

    if "rne" in what:
        Ntot = args.n_rne or (10000000 if world > 1 else 1250000)
        sb = rtbhip.ShardedBatch(Ntot, rank, world)
        N = sb.count
        rob = rtbhip.models.DH.Panda()
        ql = rob.qlim
        rng = np.random.default_rng(3 + 1000 * rank)          # rank 0 of a single-GPU run = SURVEY 8d config 4's seed
        qh = rng.uniform(ql[0], ql[1], (N, 7))
        qdh, qddh = rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
        q, qd, qdd = (torch.from_numpy(x).cuda() for x in (qh, qdh, qddh))
        hold = {}
        # the timed step is the C-ABI call on a result buffer allocated once (as bench.py does for the headline): with the Python wrapper
        # allocating 56 B x N per step, torch's caching allocator decides the average at 1e7 triples (visit w: 0.59 ms average against a
        # 0.447 ms kernel).  The wrapper's own per-step time is reported beside it.
        import ctypes as C
        tau_buf = torch.empty((N, 7), dtype=torch.float64, device=q.device)
        hold["tau"] = tau_buf
        lib = rtbhip.lib()
        dh = rob._dyn_handle()
        gc_ = np.ascontiguousarray(rob._gravity_c(None))
        ptrs = [C.c_void_p(x.data_ptr()) for x in (q, qd, qdd, tau_buf)]
        gp = gc_.ctypes.data_as(C.c_void_p)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def rne_step():
            rc = lib.rtbhip_rne(dh, ptrs[0], ptrs[1], ptrs[2], N, gp, None, ptrs[3], 1, stream)
            if rc != 0:
                raise RuntimeError(lib.rtbhip_last_error().decode())
        def rne_wrapper_step():
            hold["tau_w"] = rob.rne(q, qd, qdd)
        # sustained: the warm-up outlasts the power controller's transient (profiles/r04_rne_1e7.txt); every rank runs the same counts
        w_s, k_s = sustained_counts(rne_step)
        K, W = max(args.steps, int(rk.max_over_ranks(k_s))), max(args.warmup, int(rk.max_over_ranks(w_s)))
        elapsed, avg = rk.timed_steps(rne_step, K, W)
        _, best = ev_time(rne_step, min(args.steps, 10), 0)
        wavg, _ = ev_time(rne_wrapper_step, min(args.steps, 10), 3)
        hold.pop("tau_w", None)
        step_ms = elapsed / K * 1e3
        line = {"metric": "triples/sec (DH Panda rne)", "value": Ntot / (step_ms * 1e-3), "unit": "triples/s", "n": Ntot,
                "n_gpus": world, "scaling": "strong", "ms_per_step": step_ms, "rows_rank0": N, "steps": K, "warmup": W,
                "kernel_avg_ms": avg, "kernel_min_ms": best, "python_wrapper_ms": wavg,
                "roofline": {"bound": "hbm", "achieved": 224.0 * N / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": 224.0 * N / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": 224 * N,
                             "kernel": "k_rne<7,MDH,all-revolute,kRneSigPanda> on rank 0's rows"}}
        if N == 1250000:        # the committed PMC passes are of this shard size
            tr, src = pmc_traffic(ROOT, "r05_pmc_rne.json")
            line["roofline"]["traffic"], line["roofline"]["traffic_source"] = tr, src
        if world > 1:
            line["gather_ms"] = rk.gather_ms(hold["tau"], rows=N)      # to rank 0; ragged shards go straight into place (RCCL) / padded (gloo hook)
            line["all_gather_ms"] = rk.last_gather.get("all_gather_ms")
            line["gather"] = "gather of the (rows,7) tau shards to rank 0 (gather_ms) and to every rank (all_gather_ms), %d bytes in total; %s" % (56 * Ntot, rk.last_gather.get("via"))
            if rk.shared:
                line["devices_shared"] = True
        if rank != 0:
            line = None
        if line is not None and not args.no_cpu and world == 1:
            from oracle import ref_harness
            if ref_harness.available():
                ref = ref_harness.RefRNE(rob.L24(), 1)
                n = 100000
                t0 = time.perf_counter(); tau = ref.rne(qh[:n], qdh[:n], qddh[:n]); dt = time.perf_counter() - t0
                g = rob.rne(q[:n], qd[:n], qdd[:n]).cpu().numpy()
                line["cpu_baseline"] = {"value": n / dt, "unit": "triples/s", "cores": 1, "kind": "reference",
                                        "sample": "frne.frne per-row loop (DHRobot.rne) over the first %d triples" % n,
                                        "max_rel_err_gpu_vs_cpu": float(np.abs(g - tau).max() / np.abs(tau).max())}
        if line is not None:
            print(json.dumps(line), flush=True)

    if "dyn" in what:
        # M(q), C(q,qd), forward dynamics for the DH Panda, one fused kernel each
        N = args.n_dyn
        rob = rtbhip.models.DH.Panda()
        ql = rob.qlim
        rng = np.random.default_rng(6)
        q = torch.from_numpy(rng.uniform(ql[0], ql[1], (N, 7))).cuda()
        qd = torch.from_numpy(rng.normal(size=(N, 7))).cuda()
        tq = torch.from_numpy(rng.normal(size=(N, 7)) * 5).cuda()
        for name, fn, passes, byts in (("gravload", lambda: rob.gravload(q), 1, 56 + 56),
                                       ("inertia", lambda: rob.inertia(q), 7, 56 + 392),
                                       ("coriolis", lambda: rob.coriolis(q, qd), 7, 112 + 392),
                                       ("accel", lambda: rob.accel(q, qd, tq), 8, 168 + 56)):
            avg, best = ev_time(fn, max(3, args.steps // 2), 2)
            line = {"metric": "configurations/sec (DH Panda %s)" % name, "value": N / (avg * 1e-3), "unit": "configurations/s",
                    "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best, "rne_passes_per_config": passes,
                    "rne_passes_per_s": passes * N / (avg * 1e-3),
                    "pass_kind": {"gravload": "one pass at qd = 0: acceleration-only forward recursion, gravity as the base's acceleration (k_rne_atrest)",
                                  "inertia": "acceleration-only passes from link i on, mirrored (csrc/rne_device.h ACC)",
                                  "coriolis": "one two-field pass per column: C[:, k] = B(qd, e_k), the bilinear form of the velocity torque evaluated directly from link k on (csrc/dyn_device.h rne_bilinear_core; the reference runs 28 full passes, rounds 1-3 ran 14)",
                                  "accel": "1 full pass + 7 acceleration-only passes + LDL^T solve"}[name],
                    "roofline": valu_roofline(name, N / (avg * 1e-3), {"gravload": "k_rne_atrest<7,MDH,kRneSigPanda>", "inertia": "k_dyn<7,MDH,inertia,kRneSigPanda>",
                                                                        "coriolis": "k_dyn<7,MDH,coriolis,kRneSigPanda>", "accel": "k_dyn<7,MDH,accel,kRneSigPanda>"}[name], byts)}
            if not args.no_cpu and name == "inertia":
                from oracle import ref_harness
                if ref_harness.available():
                    ref = ref_harness.RefRNE(rob.L24(), 1)
                    n = 3000
                    qh = q[:n].cpu().numpy()
                    t0 = time.perf_counter()
                    Mc = np.array([ref.rne(np.tile(qk, (7, 1)), np.zeros((7, 7)), np.eye(7), gravity=[0, 0, 0]) for qk in qh])
                    dt = time.perf_counter() - t0
                    g = rob.inertia(q[:n]).cpu().numpy()
                    line["cpu_baseline"] = {"value": n / dt, "unit": "configurations/s", "cores": 1, "kind": "reference",
                                            "sample": "Dynamics.inertia's loop (7 frne.frne calls per configuration) over the first %d" % n,
                                            "max_abs_err_gpu_vs_cpu": float(np.abs(g - Mc).max())}
            print(json.dumps(line), flush=True)

    if "kin" in what:
        # the other kinematics outputs of the Panda chain at N = 1e6: fkine only, jacob0 only, hessian0, jacob0_dot, manipulability
        N = args.n_dyn
        ets = rtbhip.models.Panda().ets()
        rng = np.random.default_rng(0)
        q = torch.from_numpy(rng.uniform(-np.pi, np.pi, (N, 7))).cuda()
        qd = torch.from_numpy(rng.normal(size=(N, 7))).cuda()
        for name, fn, byts in (("fkine", lambda: ets.eval(q), 56 + 128), ("jacob0", lambda: ets.jacob0(q), 56 + 336),
                               ("hessian0", lambda: ets.hessian0(q), 56 + 2352), ("jacob0_dot", lambda: ets.jacob0_dot(q, qd), 112 + 336),
                               ("manipulability", lambda: ets.manipulability(q), 56 + 8), ("jacobm", lambda: ets.jacobm(q), 56 + 56)):
            extra = {}
            if name in VALU_PER_UNIT:
                # the general kernel first (what any robot gets until its own instantiation is compiled), then the robot's structure instantiation
                # (k_kin_diff<7, MODE, SIG>, compiled at run time; "jit" = 2: the launch waits for it) on the same buffers: the same bits
                rtbhip.tune("diff_sig", 0)
                g_avg, g_best = ev_time(fn, args.steps, 2)
                g_out = fn().clone()
                rtbhip.tune("diff_sig", 1); rtbhip.tune("jit", 2)
                fn(); torch.cuda.synchronize()
                rtbhip.tune("jit", 1)
                extra = {"general_kernel_avg_ms": g_avg, "general_kernel_min_ms": g_best}
            avg, best = ev_time(fn, args.steps, 2)
            if name in VALU_PER_UNIT:      # jacob0_dot / manipulability / jacobm: a few hundred bytes of I/O against 1.5-2.7 k instructions per configuration
                extra["structure_vs_general_bit_equal"] = bool(torch.equal(fn(), g_out))
                extra["structure_over_general"] = g_avg / avg
                rf = valu_roofline(name, N / (avg * 1e-3), "k_kin_diff<7,%s,SIG> (run-time structure instantiation; general kernel: general_kernel_avg_ms)" % name, byts)
            else:
                rf = {"bound": "hbm", "achieved": byts * N / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                      "frac": byts * N / (avg * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": byts * N}
            print(json.dumps({"metric": "configurations/sec (Panda %s)" % name, "value": N / (avg * 1e-3), "unit": "configurations/s", "n": N,
                              "kernel_avg_ms": avg, "kernel_min_ms": best, "roofline": rf, **extra}), flush=True)

        # fkine_all: the 8 link frames of the DH Panda (1 KB per configuration out)
        arm = rtbhip.models.DH.Panda()
        avg, best = ev_time(lambda: arm.fkine_all(q), args.steps, 2)
        byts = 56 + 8 * 128
        print(json.dumps({"metric": "configurations/sec (DH Panda fkine_all, 8 frames)", "value": N / (avg * 1e-3), "unit": "configurations/s", "n": N,
                          "kernel_avg_ms": avg, "kernel_min_ms": best,
                          "roofline": {"bound": "hbm", "achieved": byts * N / (avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                       "frac": byts * N / (avg * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": byts * N}}), flush=True)
        # partial_fkine0 order 3: (N,7,7,6,7) output = 16.5 KB per configuration, so a tenth of the batch; the time includes the
        # fkine/jacob0/hessian0 launch that feeds it and the temporaries' allocation (whole-call time, host clock around a sync)
        Np = N // 10
        qp = q[:Np].contiguous()
        ets.partial_fkine0(qp, 3); torch.cuda.synchronize()
        ts = []
        for _ in range(max(3, args.steps // 4)):
            t0 = time.perf_counter(); ets.partial_fkine0(qp, 3); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        avg = 1e3 * sum(ts) / len(ts)
        # device-side duration of the call's launches (the Jacobian + Hessian kernel that feeds it, then k_partial3): HIP events on the stream
        dev_avg, dev_min = ev_time(lambda: ets.partial_fkine0(qp, 3), max(3, args.steps // 2), 1)
        byts = 56 + 8 * 7 * 7 * 6 * 7
        print(json.dumps({"metric": "configurations/sec (Panda partial_fkine0 n=3)", "value": Np / (avg * 1e-3), "unit": "configurations/s", "n": Np,
                          "call_avg_ms": avg, "call_min_ms": 1e3 * min(ts), "kernel_avg_ms": dev_avg, "kernel_min_ms": dev_min,
                          "roofline": {"bound": "hbm", "achieved": byts * Np / (dev_avg * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                       "frac": byts * Np / (dev_avg * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_launch": byts * Np,
                                       "kernel": "k_kin_hess_tile<7,4> + k_partial3<3>: both launches of the call, device-side (the host-clock "
                                                 "time of the whole call, temporaries included, is call_avg_ms)"}}), flush=True)

    if "poe" in what:
        # north_star's "SE(3) DH / product-of-exponentials chain": a 6-joint PoE robot (UR5-like screw axes), twists lowered by
        # rtbhip_chain_create_poe to the same canonical chain -- the headline kernel class (k_kin_reg<6>), priced on the same roof
        N = args.n_dyn
        axes = [([0, 0, 1], [0, 0, 0]), ([0, 1, 0], [0, 0, 0.089]), ([0, 1, 0], [0.425, 0, 0.089]), ([0, 1, 0], [0.817, 0, 0.089]),
                ([0, 0, -1], [0.817, 0.109, 0]), ([0, 1, 0], [0.817, 0, -0.006])]
        T0 = np.array([[-1.0, 0, 0, 0.817], [0, 0, 1, 0.191], [0, 1, 0, -0.006], [0, 0, 0, 1]])
        robot = rtbhip.PoERobot([rtbhip.PoERevolute(a, p) for a, p in axes], T0, name="UR5-PoE")
        chain = robot._path(None, None)
        rng = np.random.default_rng(8)
        qh = rng.uniform(-np.pi, np.pi, (N, 6))
        q = torch.from_numpy(qh).cuda()
        hold = {}
        def poe_step():
            hold["o"] = chain.fkine_jacob0(q)
        avg, best = ev_time(poe_step, args.steps, 3)
        byts = 48 + 128 + 288
        line = {"metric": "configurations/sec (6-joint PoE robot fkine+jacob0, twists lowered by rtbhip_chain_create_poe)", "value": N / (avg * 1e-3),
                "unit": "configurations/s", "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best,
                "roofline": {"bound": "hbm", "achieved": byts * N / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": byts * N / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byts * N,
                             "kernel": "k_kin_reg<6,true,true>"}}
        if not args.no_cpu:
            from oracle import poe as opoe
            ref = opoe.PoE([opoe.unit_revolute(a, p) for a, p in axes], T0)
            n = 3000
            t0 = time.perf_counter(); Tc = ref.fkine(qh[:n]); Jc = ref.jacob0(qh[:n]); dt = time.perf_counter() - t0
            Tg, Jg = hold["o"]
            line["cpu_baseline"] = {"value": n / dt, "unit": "configurations/s", "cores": 1, "kind": "port",
                                    "sample": "closed-form PoERobot.fkine + jacob0 restated in NumPy (oracle/poe.py; the reference's own is Python over "
                                              "spatialmath, absent here) on the first %d configurations" % n,
                                    "max_abs_err_gpu_vs_cpu": max(float(np.abs(Tg[:n].cpu().numpy() - Tc).max()), float(np.abs(Jg[:n].cpu().numpy() - Jc).max()))}
        print(json.dumps(line), flush=True)

    if "graph" in what:
        # launch-bound regime (a control loop's batch): fkine+jacob0, hessian0 and rne on 4096 configurations, eager vs one captured hipGraph
        N = 4096
        ets = rtbhip.models.Panda().ets()
        arm = rtbhip.models.DH.Panda()
        rng = np.random.default_rng(0)
        q = torch.from_numpy(rng.uniform(-2, 2, (N, 7))).cuda()
        qd = torch.from_numpy(rng.normal(size=(N, 7))).cuda()
        def step():
            return ets.fkine_jacob0(q), ets.hessian0(q), arm.rne(q, qd, qd)
        step(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = step()
        reps = 300
        def clock(fn):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6
        te, tg = clock(step), clock(g.replay)
        print(json.dumps({"metric": "us per control step (Panda fkine+jacob0, hessian0, rne; N=4096)", "eager_us": te, "hipgraph_us": tg,
                          "value": N / (tg * 1e-6), "unit": "configurations/s", "n": N}), flush=True)

    if "tree" in what:
        # Robot.rne of a URDF arm (UR5, 6 link groups, <inertial> masses) through k_tree_rne
        from rtbhip import urdf
        N = args.n_dyn
        er = urdf.load("UR5").erobot()
        rng = np.random.default_rng(7)
        q, qd, qdd = (torch.from_numpy(x).cuda() for x in (rng.uniform(-3, 3, (N, er.n)), rng.normal(size=(N, er.n)), rng.normal(size=(N, er.n))))
        avg, best = ev_time(lambda: er.rne(q, qd, qdd), args.steps, 3)
        byts = 32 * er.n
        print(json.dumps({"metric": "triples/sec (URDF UR5 Robot.rne, %d link groups)" % er.n, "value": N / (avg * 1e-3), "unit": "triples/s",
                          "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best,
                          "roofline": valu_roofline("tree_ur5", N / (avg * 1e-3), "k_tree_rne<6, kTreeSigUR>", byts)}), flush=True)
        # Dynamics.gravload of the same arm: rtbhip_tree_rne with qd = NULL -> the at-rest instantiation (no velocity half, no qd row)
        avg, best = ev_time(lambda: er.gravload(q), args.steps, 3)
        print(json.dumps({"metric": "configurations/sec (URDF UR5 gravload, %d link groups)" % er.n, "value": N / (avg * 1e-3), "unit": "configurations/s",
                          "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best,
                          "roofline": valu_roofline("tree_gravload_ur5", N / (avg * 1e-3), "k_tree_rne<6, at rest, kTreeSigUR>", 16 * er.n)}), flush=True)
        # the Dynamics-mixin terms of the same URDF arm (Dynamics.inertia / coriolis / accel over Robot.rne): k_tree_dyn<6, mode>
        tq = qdd
        for name, fn, byts, key in (("inertia", lambda: er.inertia(q), 8 * er.n + 8 * er.n * er.n, "tree_inertia_ur5"),
                                    ("coriolis", lambda: er.coriolis(q, qd), 16 * er.n + 8 * er.n * er.n, "tree_coriolis_ur5"),
                                    ("accel", lambda: er.accel(q, qd, tq), 32 * er.n, "tree_accel_ur5")):
            avg, best = ev_time(fn, max(3, args.steps // 2), 2)
            line = {"metric": "configurations/sec (URDF UR5 %s, Dynamics mixin over Robot.rne)" % name, "value": N / (avg * 1e-3),
                    "unit": "configurations/s", "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best}
            if key in VALU_PER_UNIT:
                line["roofline"] = valu_roofline(key, N / (avg * 1e-3), "k_tree_dyn<6,%s,kTreeSigUR>" % name, byts)
            else:
                line["hbm_GBs"] = byts * N / (avg * 1e-3) / 1e9
            print(json.dumps(line), flush=True)

    if "ik" in what:
        N = args.n_ik
        ets = rtbhip.models.Panda().ets()
        ets.qlim = rtbhip.models.PANDA_QLIM
        rng = np.random.default_rng(1)
        qs = torch.from_numpy(rng.uniform(ets.qlim[0], ets.qlim[1], (N, 7))).cuda()
        Tep = ets.eval(qs)
        res = {}
        if world > 1:
            # BASELINE configs[2] over the ranks: the same 1e5 targets, each rank solves its row block (targets are independent;
            # imbalance between targets is handled inside each GPU).  Inside sb.ik_rows() the restart generator is keyed by the
            # GLOBAL row number (rtbhip_ik_target_base), so the gathered solutions are the single-GPU run's, whatever the split.
            sb = rtbhip.ShardedBatch(N, rank, world)
            Tl = Tep[sb.begin:sb.begin + sb.count].contiguous()
            def run_local():
                with sb.ik_rows():
                    res["out"] = ets.ik_LM(Tl, seed=2)
            w_s, k_s = sustained_counts(run_local)
            K, W = max(3, args.steps // 4, int(rk.max_over_ranks(k_s))), max(1, int(rk.max_over_ranks(w_s)))
            elapsed, dev_ms = rk.timed_steps(run_local, K, W)
            _, ok, it, _, _ = res["out"]
            solved, iters = rk.sum_over_ranks(float(ok.sum())), rk.sum_over_ranks(float(it.sum()))
            step_s = elapsed / K
            line = {"metric": "solves/sec (Panda ik_LM chan k=1, joint limits, ilimit 30 slimit 100 tol 1e-6; %d targets in row blocks over the ranks)" % N,
                    "value": N / step_s, "unit": "solves/s", "n": N, "n_gpus": world, "scaling": "strong", "ms_per_step": step_s * 1e3,
                    "rows_rank0": sb.count, "kernel_avg_ms": dev_ms, "success_rate": solved / N, "mean_iterations": iters / N,
                    "lm_iterations_per_s": iters / step_s, "roofline": ik_roofline(iters / step_s / world)}
            line["roofline"]["note"] = "per GPU: whole-job LM iterations / s divided by the world size"
            if rk.shared:
                line["devices_shared"] = True
            if rank == 0:
                print(json.dumps(line), flush=True)
            what = [w for w in what if w != "ik"]
    if "ik" in what:
        N = args.n_ik
        def run():
            res["out"] = ets.ik_LM(Tep, seed=2)
        avg, best = ev_time(run, max(3, args.steps // 4), 1)
        q, ok, it, se, E = res["out"]
        line = {"metric": "solves/sec (Panda ik_LM chan k=1, joint limits, ilimit 30 slimit 100 tol 1e-6)",
                "value": N / (avg * 1e-3), "unit": "solves/s", "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best,
                "success_rate": float(ok.float().mean()), "mean_iterations": float(it.float().mean()),
                "max_iterations": int(it.max()), "lm_iterations_per_s": float(it.sum()) / (avg * 1e-3),
                "roofline": ik_roofline(float(it.sum()) / (avg * 1e-3))}
        if not args.no_cpu:
            from oracle import ref_harness, chains
            if ref_harness.available():
                ref = ref_harness.RefETS(chains.panda_ets(with_limits=True))
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
        # SURVEY 8d config 3, second setting: the reference's ik_benchmark notebook (k = 0.1, no joint-limit rejection); and the
        # throughput regime of the default setting (10x the targets: the chip is full, the tail amortised)
        def extra(metric, T, **kw):
            r = {}
            def go():
                r["o"] = ets.ik_LM(T, seed=2, **kw)
            a, b = ev_time(go, max(3, args.steps // 4), 1)
            _, ok2, it2, _, _ = r["o"]
            n2 = T.shape[0]
            print(json.dumps({"metric": metric, "value": n2 / (a * 1e-3), "unit": "solves/s", "n": n2, "kernel_avg_ms": a, "kernel_min_ms": b,
                              "success_rate": float(ok2.float().mean()), "mean_iterations": float(it2.float().mean()),
                              "lm_iterations_per_s": float(it2.sum()) / (a * 1e-3),
                              "roofline": ik_roofline(float(it2.sum()) / (a * 1e-3))}), flush=True)
        extra("solves/sec (Panda ik_LM chan k=0.1, joint_limits=False: the ik_benchmark notebook setting)", Tep, k=0.1, joint_limits=False)
        qs10 = torch.from_numpy(rng.uniform(ets.qlim[0], ets.qlim[1], (10 * N, 7))).cuda()
        extra("solves/sec (Panda ik_LM defaults, %d targets)" % (10 * N), ets.eval(qs10))

    if "fleet" in what:
        # BASELINE configs[4]: 16 URDF arms (rtbhip/data/urdf, 4..10 joints on the path to the deepest
        # leaf), N configurations each, q ~ U(qlim) seed 4+i, ONE variable-length-chain launch
        from rtbhip import urdf
        Ntot = args.n_fleet                                    # per arm, over all ranks
        sb = rtbhip.ShardedBatch(Ntot, rank, world)
        N = sb.count
        robots = [urdf.load(nm) for nm in urdf.FLEET16]
        chs = [r.ets() for r in robots]
        qs = []
        for i, c in enumerate(chs):
            ql = np.clip(c.qlim, -2 * np.pi, 2 * np.pi)
            qs.append(torch.from_numpy(np.random.default_rng(4 + i + 1000 * rank).uniform(ql[0], ql[1], (N, c.n))).cuda())
        hold = {}
        # the legs before this one leave torch's caching allocator full of blocks of other sizes; it then serves this leg's 32 result
        # tensors per call (7.1 GB) by splitting them and falls back to hipMalloc inside the timed loop (visit r3g: 2.6 ms per step
        # against 1.45 ms of kernels; the leg alone: 1.49 ms).  Start from an empty cache; the warm-up calls size it for this leg.
        torch.cuda.empty_cache()
        hold["out"] = rtbhip.fleet_fkine_jacob(chs, qs)            # the result buffers, allocated once: the timed steps write into them
        def fleet_step():
            rtbhip.fleet_fkine_jacob(chs, qs, out=hold["out"])
        elapsed, avg = rk.timed_steps(fleet_step, max(3, args.steps // 2), 3)
        _, best = ev_time(fleet_step, 3, 0)
        step_ms = elapsed / max(3, args.steps // 2) * 1e3
        byts = sum(N * (8 * c.n + 128 + 48 * c.n) for c in chs)
        line = {"metric": "configurations/sec (mixed fleet: %d URDF arms x %d, one launch per rank)" % (len(chs), Ntot),
                "value": Ntot * len(chs) / (step_ms * 1e-3), "unit": "configurations/s", "n_gpus": world, "scaling": "strong",
                "ms_per_step": step_ms, "kernel_avg_ms": avg, "kernel_min_ms": best,
                "arms": {nm: c.n for nm, c in zip(urdf.FLEET16, chs)},
                "roofline": {"bound": "hbm", "achieved": byts / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": byts / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byts,
                             "kernel": "k_fleet<0> + k_fleet<1> on rank 0's rows"}}
        if rk.shared:
            line["devices_shared"] = True
        if world > 1:
            if rank == 0:
                print(json.dumps(line), flush=True)
            rk.finish()
            return
        # the same 16 batches through the per-chain register-resident kernel, 16 launches
        def per_chain():
            for c, q in zip(chs, qs):
                c.fkine_jacob0(q)
        avg2, best2 = ev_time(per_chain, max(3, args.steps // 2), 1)
        line["per_chain_launches"] = {"value": N * len(chs) / (avg2 * 1e-3), "kernel_avg_ms": avg2,
                                      "achieved_GBs": byts / (avg2 * 1e-3) / 1e9}
        if not args.no_cpu:
            from oracle import ref_harness
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import chain_from_ets
            if ref_harness.available():
                n = 20000
                t_cpu, err = 0.0, 0.0
                for c, q in zip(chs, qs):
                    ref = ref_harness.RefETS(chain_from_ets(c))
                    qh = q[:n].cpu().numpy()
                    t0 = time.perf_counter(); Tc = ref.fkine(qh); Jc = ref.jacob0_batch(qh); t_cpu += time.perf_counter() - t0
                    Tg, Jg = c.fkine_jacob0(q[:n])
                    err = max(err, float(np.abs(Tg.cpu().numpy() - Tc).max()), float(np.abs(Jg.cpu().numpy() - Jc).max()))
                line["cpu_baseline"] = {"value": n * len(chs) / t_cpu, "unit": "configurations/s", "cores": 1, "kind": "reference",
                                        "sample": "first %d configurations of each of the 16 arms; ETS_fkine + per-row ETS_jacob0" % n,
                                        "max_abs_err_gpu_vs_cpu": err}
        print(json.dumps(line), flush=True)
        # The 14-DOF entry of BASELINE config 5: YuMi is ONE robot with two 7-joint arms (+ finger joints).  The reference
        # evaluates a branch on the robot-wide q (Robot.jacob0(q, end=...), robot/Robot.py:1974-1981), so both arms read the
        # SAME (N, 18) array here: 17 chains of 16 robots in one launch, T + J0 for each hand.
        yumi = robots[-1]
        ends = [nm for nm in ("gripper_r_finger_r", "gripper_l_finger_l") if nm in yumi.linkdict]
        if len(ends) == 2:
            arms = [yumi.ets(end=e, compact=False) for e in ends]
            lo, hi = np.full(yumi.n, -1.0), np.full(yumi.n, 1.0)
            for a in arms:
                ql = np.clip(a.qlim, -2 * np.pi, 2 * np.pi)
                lo[a.jindices], hi[a.jindices] = ql[0], ql[1]
            qy = torch.from_numpy(np.random.default_rng(99).uniform(lo, hi, (N, yumi.n))).cuda()
            chs17, qs17 = chs[:-1] + arms, qs[:-1] + [qy, qy]
            hold.pop("out", None)
            torch.cuda.empty_cache()                                # as above: the 34 result tensors of this leg have other sizes
            hold["out17"] = rtbhip.fleet_fkine_jacob(chs17, qs17)
            def fleet17():
                rtbhip.fleet_fkine_jacob(chs17, qs17, out=hold["out17"])
            avg17, best17 = ev_time(fleet17, max(3, args.steps // 2), 3)
            byts17 = sum(N * (8 * c.n + 128 + 48 * c.n) for c in chs[:-1]) + N * (8 * yumi.n + 2 * (128 + 48 * 8))
            T17, J17 = hold["out17"]
            chk = [yumi.ets(end=e) for e in ends]                   # the same branches with path-local joint numbers
            err17 = 0.0
            for a, c, Tg, Jg in zip(arms, chk, T17[-2:], J17[-2:]):
                Tc, Jc = c.fkine_jacob0(qy[:4096][:, torch.from_numpy(a.jindices).cuda()].contiguous())
                err17 = max(err17, float((Tg[:4096] - Tc).abs().max()), float((Jg[:4096] - Jc).abs().max()))
            print(json.dumps({"metric": "configurations/sec (mixed fleet, YuMi as the 14-DOF dual-arm robot: both hands from its %d-column q; "
                                        "17 chains of 16 robots x %d, one launch)" % (yumi.n, N),
                              "value": N * 17 / (avg17 * 1e-3), "unit": "chain evaluations/s", "robots_per_s": N * 16 / (avg17 * 1e-3), "n_gpus": 1,
                              "kernel_avg_ms": avg17, "kernel_min_ms": best17, "yumi_q_columns": int(yumi.n), "yumi_arm_joints": [int(a.n) for a in arms],
                              "wide_q_vs_path_q_max_abs_diff": err17,
                              "roofline": {"bound": "hbm", "achieved": byts17 / (avg17 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": byts17 / (avg17 * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byts17}}), flush=True)
    if "servo" in what and world == 1:
        # tools/p_servo.py's error vector over N pose pairs, both methods (k_angle_axis<RPY>): 2 x 128 B in + 48 B out per pair, HBM-bound
        N = args.n_dyn
        Te = rtbhip.models.Panda().ets().eval(torch.from_numpy(np.random.default_rng(21).uniform(-np.pi, np.pi, (N, 7))).cuda())
        Tep = rtbhip.models.Panda().ets().eval(torch.from_numpy(np.random.default_rng(22).uniform(-np.pi, np.pi, (N, 7))).cuda())
        for method in ("rpy", "angle-axis"):
            hold = {}
            def servo_step():
                hold["o"] = rtbhip.p_servo(Te, Tep, method=method)
            avg, best = ev_time(servo_step, args.steps, 3)
            byts = 2 * 128 + 48
            print(json.dumps({"metric": "pose pairs/sec (p_servo, method %r: error vector, gain and arrived flag in one launch, rtbhip_p_servo)" % method,
                              "value": N / (avg * 1e-3), "unit": "pairs/s", "n": N, "kernel_avg_ms": avg, "kernel_min_ms": best,
                              "roofline": {"bound": "hbm", "achieved": byts * N / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": byts * N / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": byts * N,
                                           "kernel": "k_angle_axis<%s> (+ two torch elementwise kernels of the wrapper)" % ("true" if method == "rpy" else "false")}}), flush=True)
    rk.finish()


if __name__ == "__main__":
    main()

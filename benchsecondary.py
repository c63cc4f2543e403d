"""benchsecondary.py -- the `secondary` object of bench.py's JSON line: BASELINE.json configs[2], [3] and [4] (IK, RNE, mixed fleet)
measured by the SAME run that measures the headline, so that the driver's record carries them and nobody has to quote bench_extra.py.

Rank 0 at N = 1 only, after the timed region and outside `value` (like host_path / cpu_baseline).  Each entry:
    value / unit        whole-leg rate (units per second of the average launch)
    kernel_avg_ms       SUSTAINED device-side duration: >= 30 ms of warm-up launches, then ONE HIP-event pair on the launch stream around
                        >= 30 ms of launches / their number (benchlib.sustained_ms: the dense-fp64 kernels run at a boost clock for ~2 ms after an
                        idle gap and through a throttling transient after that, profiles/r04_rne_1e7.txt); burst_ms_after_idle = the boost figure
    parity              an in-run comparison of the GPU's output with the reference's own compiled code (oracle/_ref: fknm / frne built
                        unmodified from the reference sources; the plain-C restatement oracle/liboracle.so where _ref is absent) on a
                        bounded sample of the SAME inputs, with the tolerance it is held to -- a leg that misses it aborts the bench
    roofline            algorithmic bytes (HBM-bound legs) or algorithmic flops (IK) per launch / kernel_avg_ms against the peak;
                        the formulas are DESIGN.md section 5's
The whole object is budgeted at <= 20 s of wall time (inputs are generated on the device; the CPU samples are small).

The oracle is used here as the checker only (it is never what is timed, and nothing under robotics-toolbox-python_amd/ imports it)."""
import time

from benchlib import HBM_PEAK_GBS, FP64_VALU_PEAK_TFLOPS, sustained_ms

# fp64 operations of ONE Levenberg-Marquardt iteration of the 7-joint Panda as the ALGORITHM needs them (fused multiply-add = 2):
# FK + Jacobian walk incl. 7 sincos ~0.60 k, angle-axis error + E ~0.12 k, J^T W J + g (lower triangle) ~0.46 k, 7x7 LDL^T factor + solves
# ~0.25 k, update / wrap / limit test ~0.05 k.  (The reference's own formulation spends ~5.5 k)
IK_FLOPS_PER_ITERATION = 1480.0
RNE_BYTES_PER_TRIPLE = 224          # 3 x 56 B read + 56 B written


def _burst_ms(fn, reps=3):
    """Average of the first `reps` launches after an idle gap (the boost clock; reported beside the sustained figure, never as the number)."""
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _hbm(bytes_per_launch, ms, kernel):
    a = bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": int(bytes_per_launch), "kernel": kernel, "traffic": None}


def _oracle_chain(ets):
    """The checker's chain with the same op-table as a product ETS (joints numbered 0..n-1 in order of appearance)."""
    import numpy as np
    from oracle import chains
    spec = []
    for kind, flip, jindex, T in ets.optable():
        spec.append(np.array(T, dtype=np.float64) if kind == 6 else (("Rx", "Ry", "Rz", "tx", "ty", "tz")[kind], None, bool(flip)))
    return chains.Chain(spec, qlim=ets._limits(False))


def _kin_checker(ch):
    """(kind, fkine, jacob0) of the reference's compiled fknm for this chain, else of the C restatement."""
    try:
        from oracle import ref_harness
        if not ref_harness.available():
            raise ImportError("oracle/_ref not built")
        ref = ref_harness.RefETS(ch)
        return "reference", ref, ref.fkine, ref.jacob0_batch
    except Exception:
        from oracle import oracle
        return "port", None, (lambda a: oracle.fkine(ch, a)), (lambda a: oracle.jacob0(ch, a))


def ik_config3(rtbhip, N=100000, sample=2000):
    """BASELINE configs[2]: ik_LM over 1e5 random reachable targets, Franka limits, the defaults (chan, k = 1, ilimit 30, slimit 100,
    tol 1e-6), restarts on the device, seed 2.  Parity: `sample` of the same targets solved from a supplied q0 (no generator involved:
    SURVEY 8c) by the reference's IK_LM_c and by the GPU -- wherever the reference converges in its first search, (success, iterations,
    searches) must be EQUAL and q within 1e-6."""
    import numpy as np
    import torch
    from oracle import chains
    ets = rtbhip.models.Panda().ets()
    ets.qlim = rtbhip.models.PANDA_QLIM
    rng = np.random.default_rng(1)
    qs_h = rng.uniform(ets.qlim[0], ets.qlim[1], (N, 7))
    qs = torch.from_numpy(qs_h).cuda()
    Tep = ets.eval(qs)
    res = {}

    def run():
        res["out"] = ets.ik_LM(Tep, seed=2)
    run()
    burst = _burst_ms(run)
    ms, reps, warm = sustained_ms(run)
    q, ok, it, se, E = res["out"]
    okb = ok.bool()
    its = float(it.sum())
    lm_per_s = its / (ms * 1e-3)
    tf = lm_per_s * IK_FLOPS_PER_ITERATION / 1e12
    # every reported success is a solution: E < tol as reported, joint limits respected
    lim = torch.from_numpy(np.asarray(ets.qlim)).cuda()
    assert bool((E[okb] < 1e-6).all()) and bool(((q[okb] >= lim[0]) & (q[okb] <= lim[1])).all()), "ik: a reported success is not one"
    out = {"workload": "BASELINE configs[2]: ETS Panda with the Franka limits, %d targets Tep = FK(q*), q* ~ U(qlim) seed 1; ik_LM defaults "
                       "(chan, k=1, ilimit 30, slimit 100, tol 1e-6, joint limits), restart seed 2" % N,
           "value": N / (ms * 1e-3), "unit": "solves/s", "n": N, "kernel_avg_ms": ms, "launches_timed": reps, "launches_warmup": warm,
           "burst_ms_after_idle": burst, "success_rate": float(okb.float().mean()), "mean_iterations": its / N, "lm_iterations_per_s": lm_per_s,
           "roofline": {"bound": "fp64-valu", "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
                        "flops_per_iteration": IK_FLOPS_PER_ITERATION, "kernel": "k_ik<7,0>",
                        "counts": "only the iterations the reference's sequential loop reports (discarded speculation and idle lanes count against the kernel)"}}
    # parity against the reference's IK_LM_c on a sample, q0 supplied
    n = min(sample, N)
    ch = chains.panda_ets(with_limits=True)
    kind, ref, _, _ = _kin_checker(ch)
    Th = Tep[:n].cpu().numpy()
    q0 = np.clip(qs_h[:n] + 0.15 * np.random.default_rng(5).normal(size=(n, 7)), ets.qlim[0] + 1e-3, ets.qlim[1] - 1e-3)
    gq, gok, git, gse, _ = ets.ik_LM(Th, q0=q0, seed=2)
    first, dq, same = 0, 0.0, 0
    t0 = time.perf_counter()
    for i in range(n):
        if ref is not None:
            o = ref.ik_LM(Th[i], q0=q0[i].copy())
            rq, rok, rit, rse = np.asarray(o[0]), int(o[1]), int(o[2]), int(o[3])
        else:
            from oracle import oracle
            o = oracle.ik_lm(ch, Th[i], q0=q0[i], restarts=np.zeros((2, 7)), slimit=1)
            rq, rok, rit, rse = np.asarray(o[0]), int(o[1]), int(o[2]), int(o[3])
        if rok and rse == 1:
            first += 1
            same += int((rok, rit, rse) == (int(gok[i]), int(git[i]), int(gse[i])))
            dq = max(dq, float(np.abs(rq - gq[i]).max()))
    cpu_s = time.perf_counter() - t0
    out["parity"] = {"against": "IK_LM_c of the reference's compiled fknm (oracle/_ref)" if kind == "reference" else "oracle/liboracle.so (C restatement of ik.cpp)",
                     "sample": "the first %d targets from a supplied q0 = q* + N(0, 0.15): rows the reference solves in its first search" % n,
                     "first_search_rows": first, "same_success_iterations_searches": same, "max_abs_dq": dq, "tolerance": 1e-6,
                     "cpu_seconds": cpu_s}
    if first < n // 2 or same != first or not dq <= 1e-6:
        raise SystemExit("bench: secondary ik parity failed: %r" % (out["parity"],))
    out["loss_factors"] = ik_loss_factors(run, its, ms)
    return out


def ik_loss_factors(run, its, ms):
    """Where k_ik's distance from the fp64 roof goes (config 3), measured by this run where it can be:
        lane_running   share of the lane slots of the kernel's LM iterations that held a running search (the rest: idle or parked lanes)
        lane_useful    share that produced an iteration the reference's sequential loop REPORTS (running minus discarded speculation:
                       searches started ahead of an earlier one that then succeeded)
        clock_ghz      effective shader clock of the waves while they ran: s_memtime cycles / s_memrealtime 100 MHz ticks, summed over waves
    from ONE extra launch of the same call with the kernel's diagnostic counters on (RTBHIP_IK_STATS: the counting instantiation of the SAME
    signature kernel, csrc/ik_kernels.hip kIkAuxStats -- six words per wave), and, from the committed SQ-counter pass of this round
    (profiles/r06_ik_sq.json: rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES over scripts/ik_loss_factors.py -- counters are their own run, never the timed one):
        valu_per_iteration   VALU instructions a wave executes per LM iteration of its loop, scheduling passes amortised in
        valu_issue_util      those instructions against the fp64 issue slots the kernel's duration offers (one wave64 instruction per SIMD per 4 cycles
                             at clock_ghz): how full the pipes are -- a diagnostic, a kernel executing more instructions for the same answer scores higher
    The product  lane_useful x (algorithmic FMAs / valu_per_iteration) x valu_issue_util x (clock_ghz / 2.4)  is the roofline fraction, up to the
    share of the flop count that is not FMA."""
    import json
    import os
    import tempfile
    import numpy as np
    import torch
    import rtbhip
    fd, path = tempfile.mkstemp(suffix=".jsonl")
    os.close(fd)
    os.environ["RTBHIP_IK_STATS"] = path
    try:
        run()
        torch.cuda.synchronize()
    finally:
        del os.environ["RTBHIP_IK_STATS"]
    rows = [json.loads(l) for l in open(path) if l.strip()]
    os.unlink(path)
    if not rows:
        return {"error": "no counters came back (RTBHIP_IK_STATS is served for the 7-joint arm only)"}
    w = np.array([x for r in rows for x in r["per_wave"]], dtype=np.float64)
    wave_iters, passes, running, cyc, ticks = w[:, 0].sum(), w[:, 1].sum(), w[:, 2].sum(), w[:, 4].sum(), w[:, 5].sum()
    clock = 0.1 * cyc / ticks if ticks > 0 else None
    out = {"wave_iterations": int(wave_iters), "scheduling_passes": int(passes), "waves": int(len(w)), "lane_running": running / (64.0 * wave_iters),
           "lane_useful": its / (64.0 * wave_iters), "clock_ghz": clock,
           "how": "one extra launch with the diagnostic counters of the same signature kernel (not timed); clock = s_memtime cycles / s_memrealtime ticks over the waves' lives"}
    here = os.path.dirname(os.path.abspath(__file__))
    sq = os.path.join(here, "profiles", "r06_ik_sq.json")
    if os.path.exists(sq):
        c = json.load(open(sq))
        vpi = c["valu_per_wave_iteration"]
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        slots = ms * 1e-3 * (clock or 2.4) * 1e9 / 4.0 * cus * 4
        out.update({"valu_per_iteration": vpi, "valu_issue_util": vpi * wave_iters / slots, "algorithmic_fma_per_iteration": IK_FLOPS_PER_ITERATION / 2.0,
                    "valu_source": "committed constant: profiles/r06_ik_sq.json (SQ_INSTS_VALU / wave iterations of the counted launches, %s)" % c.get("visit", "")})
    return out


def rne_config4(rtbhip, N=10000000, shard=1250000, sample=20000):
    """BASELINE configs[3]: DH Panda (modified DH, masses + inertia tensors) inverse dynamics over 1e7 (q, qd, qdd) triples -- the whole
    configuration on ONE GPU (2.24 GB of traffic per launch), and the 1.25e6-triple share one GPU of eight owns.  Parity: a strided sample
    of the same rows through the reference's compiled frne, relative to max |tau|."""
    import ctypes as C
    import numpy as np
    import torch
    rob = rtbhip.models.DH.Panda()
    ql = torch.from_numpy(np.asarray(rob.qlim)).cuda()
    g = torch.Generator(device="cuda").manual_seed(3)
    q = ql[0] + (ql[1] - ql[0]) * torch.rand((N, 7), dtype=torch.float64, device="cuda", generator=g)
    qd = torch.randn((N, 7), dtype=torch.float64, device="cuda", generator=g)
    qdd = torch.randn((N, 7), dtype=torch.float64, device="cuda", generator=g)
    tau = torch.empty((N, 7), dtype=torch.float64, device="cuda")
    lib = rtbhip.lib()
    dh = rob._dyn_handle()
    grav = np.ascontiguousarray(rob._gravity_c(None))
    gp = grav.ctypes.data_as(C.c_void_p)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = [C.c_void_p(x.data_ptr()) for x in (q, qd, qdd, tau)]

    def launch(n):
        def f():
            rc = lib.rtbhip_rne(dh, ptr[0], ptr[1], ptr[2], n, gp, None, ptr[3], 1, stream)
            if rc != 0:
                raise RuntimeError(lib.rtbhip_last_error().decode())
        return f
    f = launch(N)
    f()
    burst_full = _burst_ms(f)
    ms_full, reps, warm = sustained_ms(f)
    # per-launch durations in the steady state (event pairs, straight after the loop above): avg / min at this size
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    each = sorted(a.elapsed_time(b) for a, b in ev)
    # parity: strided rows of the 1e7 batch
    sel = torch.arange(0, N, max(1, N // sample), device="cuda")[:sample]
    qh, qdh, qddh, got = (x[sel].cpu().numpy() for x in (q, qd, qdd, tau))
    t0 = time.perf_counter()
    try:
        from oracle import ref_harness
        if not ref_harness.available():
            raise ImportError("oracle/_ref not built")
        ref = ref_harness.RefRNE(rob.L24(), 1)
        want = ref.rne(qh, qdh, qddh)
        against = "frne.frne of the reference's compiled extension (oracle/_ref), the per-row loop of DHRobot.rne"
    except ImportError:
        from oracle import oracle, chains
        tab = chains.panda_dh()
        want = oracle.rne_dh(tab.L24(), 1, qh, qdh, qddh, -tab.gravity)
        against = "oracle/liboracle.so (C restatement of ne.c)"
    cpu_s = time.perf_counter() - t0
    rel = float(np.abs(got - want).max() / np.abs(want).max())
    if not rel <= 1e-9:
        raise SystemExit("bench: secondary rne parity failed: rel err %g" % rel)
    fs = launch(shard)
    fs()
    burst_shard = _burst_ms(fs, 10)
    ms_shard, reps_s, warm_s = sustained_ms(fs)
    full = {"workload": "BASELINE configs[3] whole: DH Panda rne, %d triples, q ~ U(qlim), qd, qdd ~ N(0,1) (device generator seed 3), gravity [0,0,-9.81]" % N,
            "value": N / (ms_full * 1e-3), "unit": "triples/s", "n": N, "kernel_avg_ms": ms_full, "launches_timed": reps, "launches_warmup": warm,
            "burst_ms_after_idle": burst_full,
            "per_launch_event_ms": {"min": each[0], "median": each[len(each) // 2], "max": each[-1],
                                    "note": "40 launches in the steady state, one event pair each (a pair adds a few microseconds); the clock, not the "
                                            "memory system, moves these: profiles/r04_rne_1e7.txt"},
            "parity": {"against": against, "sample": "%d rows, every %d-th of the batch" % (len(got), max(1, N // sample)),
                       "max_rel_err": rel, "tolerance": 1e-9, "cpu_seconds": cpu_s, "cpu_triples_per_s": len(got) / cpu_s},
            "roofline": _hbm(RNE_BYTES_PER_TRIPLE * N, ms_full, "k_rne<7,MDH,all-revolute,kRneSigPanda>")}
    part = {"workload": "BASELINE configs[3] per-GPU share: the first %d of the same triples (what one of 8 ranks owns)" % shard,
            "value": shard / (ms_shard * 1e-3), "unit": "triples/s", "n": shard, "kernel_avg_ms": ms_shard, "launches_timed": reps_s,
            "launches_warmup": warm_s, "burst_ms_after_idle": burst_shard,
            "parity": "rows of the same buffers and the same kernel as rne_config4_1e7 (its sample covers this range)",
            "roofline": _hbm(RNE_BYTES_PER_TRIPLE * shard, ms_shard, "k_rne<7,MDH,all-revolute,kRneSigPanda>")}
    del q, qd, qdd, tau
    torch.cuda.empty_cache()
    return full, part


def fleet_config5(rtbhip, N=1000000, sample=4000):
    """BASELINE configs[4] in full: the 16 supplied URDF arms, N configurations each, q ~ U(qlim) (device generator, seed 4 + i), ONE
    variable-length-chain call for all of them -- with YuMi as what it is, ONE 14-DOF dual-arm robot: its two arms (7 joints + a finger each)
    are two chains that read the SAME (N, 18) robot-wide q array (the reference evaluates a branch on the robot's q: Robot.jacob0(q, end=...),
    robot/Robot.py:1974-1981), so the call walks 17 chains of 16 robots and the robots span 4..14 DOF.  Timed in the two-array layout (the faster
    one: profiles/r05_layout.txt) and in the packed layout (rtbhip_fleet_fkine_jacob_packed: one [T | J] array per chain; its rows must equal the
    two-array form's bit for bit); the 16-chain form of rounds 1-4 (YuMi as one 8-joint branch) is kept beside it.  Parity: the first `sample` rows of EVERY chain through the reference's compiled ETS_fkine / ETS_jacob0."""
    import numpy as np
    import torch
    from rtbhip import urdf
    robots = [urdf.load(nm) for nm in urdf.FLEET16]
    chs16 = [r.ets() for r in robots]
    qs16 = []
    for i, c in enumerate(chs16):
        ql = torch.from_numpy(np.clip(c.qlim, -2 * np.pi, 2 * np.pi)).cuda()
        g = torch.Generator(device="cuda").manual_seed(4 + i)
        qs16.append(ql[0] + (ql[1] - ql[0]) * torch.rand((N, c.n), dtype=torch.float64, device="cuda", generator=g))
    yumi = robots[-1]
    ends = ("gripper_r_finger_r", "gripper_l_finger_l")
    arms = [yumi.ets(end=e, compact=False) for e in ends]                    # robot-wide joint numbers: both read the (N, 18) array
    lo, hi = np.full(yumi.n, -1.0), np.full(yumi.n, 1.0)
    for a in arms:
        ql = np.clip(a.qlim, -2 * np.pi, 2 * np.pi)
        lo[a.jindices], hi[a.jindices] = ql[0], ql[1]
    g = torch.Generator(device="cuda").manual_seed(4 + 15)
    lo_d, hi_d = torch.from_numpy(lo).cuda(), torch.from_numpy(hi).cuda()
    qy = lo_d + (hi_d - lo_d) * torch.rand((N, yumi.n), dtype=torch.float64, device="cuda", generator=g)
    chs17, qs17 = chs16[:-1] + arms, qs16[:-1] + [qy, qy]
    torch.cuda.empty_cache()

    def measure(chs, qs, keep=False):
        """sustained ms of the call in both layouts on preallocated outputs; returns (ms_two, ms_packed, the two-array outputs' first `sample` rows
        re-packed as [T | J] on the host when keep, timed-launch counts)"""
        hold = {"packed": rtbhip.fleet_fkine_jacob_packed(chs, qs)}

        def pk():
            rtbhip.fleet_fkine_jacob_packed(chs, qs, out=hold["packed"])
        pk()
        ms_p, _, _ = sustained_ms(pk)
        head_p = [x[:min(sample, N)].cpu().numpy() for x in hold["packed"]] if keep else None
        del hold["packed"]
        torch.cuda.empty_cache()
        hold["two"] = rtbhip.fleet_fkine_jacob(chs, qs)

        def two():
            rtbhip.fleet_fkine_jacob(chs, qs, out=hold["two"])
        two()
        ms_two, reps, warm = sustained_ms(two)
        head = None
        if keep:
            n_ = min(sample, N)
            head = [np.concatenate([T[:n_].reshape(n_, 16).cpu().numpy(), J[:n_].reshape(n_, -1).cpu().numpy()], axis=1) for T, J in zip(*hold["two"])]
            for a, b in zip(head, head_p):
                if not np.array_equal(a, b):
                    raise SystemExit("bench: the packed fleet rows differ from the two-array form")
        return ms_two, ms_p, head, reps, warm
    ms16, ms16_packed, _, _, _ = measure(chs16, qs16)
    torch.cuda.empty_cache()
    ms, ms_packed, rows, reps, warm = measure(chs17, qs17, keep=True)
    byts16 = sum(N * (8 * c.n + 128 + 48 * c.n) for c in chs16)
    byts = sum(N * (8 * c.n + 128 + 48 * c.n) for c in chs16[:-1]) + N * (8 * yumi.n + sum(128 + 48 * a.n for a in arms))     # YuMi's q row is read once per arm chain but is ONE array: priced once
    err, cpu_s, kind = 0.0, 0.0, None
    per_chain = {}
    n = min(sample, N)
    names = list(urdf.FLEET16[:-1]) + ["YuMi:" + e for e in ends]
    for nm, c, qd_, TJ in zip(names, chs17, qs17, rows):
        kind, _, fk, jc = _kin_checker(_oracle_chain(c))
        qh = qd_[:n].cpu().numpy()
        if c.q_width != c.n:
            qh = np.ascontiguousarray(qh[:, np.asarray(c.jindices)])         # the checker's chain numbers the path's joints 0..n-1 in order of appearance
        t0 = time.perf_counter()
        Tc, Jc = fk(qh), jc(qh)
        cpu_s += time.perf_counter() - t0
        got = TJ[:n]
        e = max(float(np.abs(got[:, :16].reshape(n, 4, 4) - Tc).max()), float(np.abs(got[:, 16:].reshape(n, 6, c.n) - Jc).max()))
        per_chain[nm] = e
        err = max(err, e)
    if not err <= 1e-10:
        raise SystemExit("bench: secondary fleet parity failed: max abs err %g (%r)" % (err, per_chain))
    dof = {nm: int(c.n) for nm, c in zip(urdf.FLEET16[:-1], chs16[:-1])}
    dof["YuMi"] = 14
    out = {"workload": "BASELINE configs[4]: 16 URDF arms x %d configurations (4..14 DOF; YuMi as ONE 14-DOF dual-arm robot: two 8-joint chains -- 7 arm joints + a finger "
                       "each -- reading its %d-column q), q ~ U(qlim) (device generator seed 4+i), fkine + jacob0 of every chain, ONE call (17 chains)" % (N, yumi.n),
           "value": N * 16 / (ms * 1e-3), "unit": "robot configurations/s", "n": N * 16, "chains": 17, "chain_evaluations_per_s": N * 17 / (ms * 1e-3),
           "kernel_avg_ms": ms, "layout": "two arrays per chain (T, J0)", "packed_layout_ms": ms_packed, "packed_rows_equal_two_arrays": True,
           "launches_timed": reps, "launches_warmup": warm,
           "arms": dof, "chain_joints": {nm: int(c.n) for nm, c in zip(names, chs17)},
           "parity": {"against": "ETS_fkine + per-row ETS_jacob0 of the reference's compiled fknm (oracle/_ref)" if kind == "reference" else "oracle/liboracle.so",
                      "sample": "the first %d configurations of each of the 17 chains (YuMi's two arms on the columns of its robot-wide q)" % n,
                      "max_abs_err": err, "max_abs_err_yumi_arms": max(per_chain[k] for k in per_chain if k.startswith("YuMi:")), "tolerance": 1e-10,
                      "cpu_seconds": cpu_s, "cpu_configurations_per_s": n * 17 / cpu_s},
           "roofline": _hbm(byts, ms, "k_fleet<0> + k_fleet<1> (one call, two launches: chains of up to 8 joints / beyond)"),
           "sixteen_chain_form": {"what": "rounds 1-4's form: YuMi as one 8-joint branch with path-local q (16 chains, 4..10 joints)", "n": N * 16,
                                  "kernel_avg_ms": ms16, "packed_layout_ms": ms16_packed, "value": N * 16 / (ms16 * 1e-3),
                                  "roofline": _hbm(byts16, ms16, "k_fleet<0> + k_fleet<1>"),
                                  "roofline_packed": _hbm(byts16, ms16_packed, "k_fleet<0,packed> + k_fleet<1,packed>")}}
    out["roofline"]["packed_layout_frac"] = byts / (ms_packed * 1e-3) / 1e9 / HBM_PEAK_GBS
    del rows, qs16, qs17, qy
    torch.cuda.empty_cache()
    return out


def structure_legs(rtbhip, n_rne=10000000, n_ik=100000, n_tree=1000000):
    """What a user's OWN robot gets (round-5 review, missing 2 / 3).  The fast forms of k_ik / k_rne / k_tree_* are instantiations per robot
    structure: built into the library for the robots of BASELINE's configs, compiled at run time by hipRTC (csrc/jit.cpp) for every other
    robot.  Per family, on the SAME workload shape as the config it belongs to:
        general     the general kernel (rtbhip_tune *_sig = 0) on the config's own robot -- what every robot got before this round, and what a
                    robot gets while its code object is being compiled, or on a box without hipRTC
        jit         a robot with NO built-in instantiation, served by its run-time instantiation, and the same robot on the general kernel;
                    in-run check: the two outputs are EQUAL, bit for bit (the structured forms are the general operation sequences with their
                    exact zeros and ones rewritten away: csrc/exactform.h)
    plus the compile seconds / disk-cache hits this process saw (rtbhip_jit_stats)."""
    import numpy as np
    import torch
    from rtbhip import jit, urdf
    out = {}
    if not jit.stats()["available"]:
        return {"error": "libhiprtc.so not found on this box: every robot without a built-in instantiation takes the general kernels"}

    def timed(fn):
        fn()
        ms, reps, warm = sustained_ms(fn)
        return ms

    def settle(prepare_obj):
        t0 = time.perf_counter()
        jit.prepare(prepare_obj)
        jit.wait(300)
        return time.perf_counter() - t0

    def runtime_instead_of_builtin(obj, run, res, ms_builtin):
        """The SAME robot and workload served by the kernel hipRTC compiles for its structure instead of the instantiation built into the library
        (rtbhip_tune("sig_builtin", 0)): what "compiled at run time" costs against "instantiated by hand" -- same sources, same words, another moment."""
        def snap():
            return [x.clone() for x in (res["out"] if "out" in res else [res["tau"]])]
        run()
        ref = snap()
        rtbhip.tune("sig_builtin", 0)
        try:
            wait_s = settle(obj)
            l0 = jit.stats()["launches"]
            ms = timed(run)
            served = jit.stats()["launches"] - l0
            same = all(bool(torch.equal(a, b)) for a, b in zip(ref, snap()))
        finally:
            rtbhip.tune("sig_builtin", 1)
        ms_again = timed(run)          # the built-in kernel once more, with the buffers where they are now (the comparison that counts: same placement, back to back)
        r = {"builtin_ms": ms_again, "runtime_ms": ms, "runtime_over_builtin": ms / ms_again, "builtin_ms_before_the_snapshot": ms_builtin,
             "launches_served_by_jit": int(served), "bit_identical": same, "wait_for_compile_s": wait_s}
        if not same or served < 1:
            raise SystemExit("bench: a robot's run-time instantiation differs from its built-in one (or did not serve): %r" % (r,))
        return r

    g = torch.Generator(device="cuda").manual_seed(11)

    # ---- RNE (config 4's shape): the DH Panda on the general kernel; a DH Panda with one alpha perturbed (no built-in instantiation)
    def rne_leg(rob, N):
        ql = torch.from_numpy(np.asarray(rob.qlim)).cuda()
        q = ql[0] + (ql[1] - ql[0]) * torch.rand((N, rob.n), dtype=torch.float64, device="cuda", generator=g)
        qd = torch.randn((N, rob.n), dtype=torch.float64, device="cuda", generator=g)
        qdd = torch.randn((N, rob.n), dtype=torch.float64, device="cuda", generator=g)
        res = {}

        def run():
            res["tau"] = rob.rne(q, qd, qdd)
        return run, res
    panda = rtbhip.models.DH.Panda()
    run, res = rne_leg(panda, n_rne)
    rtbhip.tune("rne_sig", 0)
    try:
        ms_gen = timed(run)
    finally:
        rtbhip.tune("rne_sig", 1)
    ms_sig = timed(run)
    out["rne_general"] = {"workload": "config 4's shape: DH Panda rne, %d triples, GENERAL kernel k_rne<7,MDH,all-revolute> (rne_sig = 0)" % n_rne,
                          "kernel_avg_ms": ms_gen, "builtin_signature_ms": ms_sig, "roofline": _hbm(RNE_BYTES_PER_TRIPLE * n_rne, ms_gen, "k_rne<7,true,true,0>")}
    out["rne_general"]["runtime_vs_builtin"] = runtime_instead_of_builtin(panda, run, res, ms_sig)
    links = list(panda.links)
    k = links[3]
    links[3] = rtbhip.RevoluteMDH(a=k.a, d=k.d, alpha=k.alpha + 0.01, m=k.m, r=k.r, I=k.I, G=1)
    other = rtbhip.DHRobot(links, name="Panda, alpha_4 + 0.01")
    wait_s = settle(other)
    run, res = rne_leg(other, n_rne)
    l0 = jit.stats()["launches"]
    ms_jit = timed(run)
    served = jit.stats()["launches"] - l0
    a = res["tau"].clone()
    rtbhip.tune("rne_sig", 0)
    try:
        ms_gen2 = timed(run)
    finally:
        rtbhip.tune("rne_sig", 1)
    same = bool(torch.equal(a, res["tau"]))
    out["rne_jit"] = {"workload": "the same, a DH Panda whose alpha_4 is 0.01 rad off: no built-in instantiation -> k_rne<7,true,true,SIG> compiled by hipRTC",
                      "kernel_avg_ms": ms_jit, "general_kernel_ms": ms_gen2, "speedup_vs_general": ms_gen2 / ms_jit, "launches_served_by_jit": int(served),
                      "bit_identical_to_general": same, "wait_for_compile_s": wait_s, "roofline": _hbm(RNE_BYTES_PER_TRIPLE * n_rne, ms_jit, jit.names(other)[0][0])}
    if not same or served < 1:
        raise SystemExit("bench: the run-time instantiation of k_rne is not the general kernel's bits (or did not serve): %r" % (out["rne_jit"],))
    del res, a
    torch.cuda.empty_cache()

    # ---- IK (config 3's shape): the Panda on the general kernel; the LBR iiwa (URDF, 7 joints, one negative axis) on its run-time instantiation
    def ik_leg(ets, N, seed):
        rng = np.random.default_rng(seed)
        lim = np.asarray(ets.qlim)
        Tep = ets.eval(torch.from_numpy(rng.uniform(lim[0], lim[1], (N, ets.n))).cuda())
        res = {}

        def run():
            res["out"] = ets.ik_LM(Tep, seed=2)
        return run, res
    pe = rtbhip.models.Panda().ets()
    pe.qlim = rtbhip.models.PANDA_QLIM
    run, res = ik_leg(pe, n_ik, 1)
    rtbhip.tune("ik_sig", 0)
    try:
        ms_gen = timed(run)
    finally:
        rtbhip.tune("ik_sig", 1)
    its = float(res["out"][2].sum())
    out["ik_general"] = {"workload": "config 3 exactly, GENERAL kernel k_ik<7,0,13,0> (ik_sig = 0)", "kernel_avg_ms": ms_gen,
                         "roofline": {"bound": "fp64-valu", "unit": "TFLOP/s", "peak": FP64_VALU_PEAK_TFLOPS,
                                      "achieved": its / (ms_gen * 1e-3) * IK_FLOPS_PER_ITERATION / 1e12,
                                      "frac": its / (ms_gen * 1e-3) * IK_FLOPS_PER_ITERATION / 1e12 / FP64_VALU_PEAK_TFLOPS, "kernel": "k_ik<7,0,13,0>"}}
    ms_sig = timed(run)
    out["ik_general"]["builtin_signature_ms"] = ms_sig
    out["ik_general"]["runtime_vs_builtin"] = runtime_instead_of_builtin(pe, run, res, ms_sig)
    lbr = urdf.load("LBR").ets()
    wait_s = settle(lbr)
    run, res = ik_leg(lbr, n_ik, 4)
    l0 = jit.stats()["launches"]
    ms_jit = timed(run)
    served = jit.stats()["launches"] - l0
    a = [x.clone() for x in res["out"]]
    rtbhip.tune("ik_sig", 0)
    try:
        ms_gen2 = timed(run)
    finally:
        rtbhip.tune("ik_sig", 1)
    counts_same = all(bool(torch.equal(a[k], res["out"][k])) for k in (1, 2, 3))
    okb = a[1].bool()
    dq = float((a[0][okb] - res["out"][0][okb]).abs().max()) if bool(okb.any()) else 0.0
    same = counts_same and dq == 0.0             # the same bits, q included (round 6: by construction, csrc/kin_device.h mix_pp)
    its = float(a[2].sum())
    out["ik_jit"] = {"workload": "config 3's shape on the LBR iiwa read from its URDF (7 joints, one negative axis, URDF limits): %d reachable targets, ik_LM defaults" % n_ik,
                     "kernel_avg_ms": ms_jit, "general_kernel_ms": ms_gen2, "speedup_vs_general": ms_gen2 / ms_jit, "launches_served_by_jit": int(served),
                     "decisions_and_counts_equal_to_general": counts_same, "max_abs_dq_vs_general": dq, "success_rate": float(a[1].bool().float().mean()), "mean_iterations": its / n_ik,
                     "wait_for_compile_s": wait_s, "kernel": (jit.names(lbr)[0] or ["?"])[0]}
    if not same or served < 1:
        raise SystemExit("bench: the run-time instantiation of k_ik is not the general kernel's bits (or did not serve): %r" % (out["ik_jit"],))
    del res, a
    torch.cuda.empty_cache()

    # ---- link trees (Robot.rne): the UR5 on the general kernel; the Kinova Gen3 (13 link groups) and YuMi (18) on their run-time instantiations
    def tree_leg(t, N):
        q = 4.0 * torch.rand((N, t.n), dtype=torch.float64, device="cuda", generator=g) - 2.0
        qd = torch.randn((N, t.n), dtype=torch.float64, device="cuda", generator=g)
        qdd = torch.randn((N, t.n), dtype=torch.float64, device="cuda", generator=g)
        res = {}

        def run():
            res["tau"] = t.rne(q, qd, qdd)
        return run, res
    ur = urdf.load("UR5").erobot()
    run, res = tree_leg(ur, n_tree)
    rtbhip.tune("tree_sig", 0)
    try:
        ms_gen = timed(run)
    finally:
        rtbhip.tune("tree_sig", 1)
    ms_sig = timed(run)
    out["tree_general"] = {"workload": "UR5 link tree (6 groups) Robot.rne, %d triples, GENERAL kernel k_tree_rne<6,false> (tree_sig = 0)" % n_tree,
                           "kernel_avg_ms": ms_gen, "builtin_signature_ms": ms_sig}
    out["tree_general"]["runtime_vs_builtin"] = runtime_instead_of_builtin(ur, run, res, ms_sig)
    for name in ("KinovaGen3", "YuMi"):
        t = urdf.load(name).erobot()
        wait_s = settle(t)
        run, res = tree_leg(t, n_tree)
        l0 = jit.stats()["launches"]
        ms_jit = timed(run)
        served = jit.stats()["launches"] - l0
        a = res["tau"].clone()
        rtbhip.tune("tree_sig", 0)
        try:
            ms_gen2 = timed(run)
        finally:
            rtbhip.tune("tree_sig", 1)
        same = bool(torch.equal(a, res["tau"]))
        out["tree_jit_" + name] = {"workload": "%s link tree (%d groups) Robot.rne, %d triples: its generated knowledge type, k_tree_rne compiled by hipRTC" % (name, t.n, n_tree),
                                   "kernel_avg_ms": ms_jit, "general_kernel_ms": ms_gen2, "speedup_vs_general": ms_gen2 / ms_jit,
                                   "launches_served_by_jit": int(served), "bit_identical_to_general": same, "wait_for_compile_s": wait_s}
        if not same or served < 1:
            raise SystemExit("bench: the run-time instantiation of k_tree_rne (%s) is not the general kernel's bits (or did not serve): %r" % (name, out["tree_jit_" + name]))
        del res, a
        torch.cuda.empty_cache()
    st = jit.stats()
    out["jit"] = {k: st[k] for k in ("requested", "compiled", "disk_hits", "failed", "launches", "general_while_pending", "compile_seconds", "compile_seconds_max", "source_digest")}
    return out


# the instantiations that serve configs 3 and 4 (the structure legs launch the general kernels and run-time instantiations of the same templates on the
# same grids: the committed profile is read for exactly these names) -- csrc/ik_kernels.hip kIkSigPandaETS, csrc/rne_device.h kRneSigPanda
K_IK_PANDA = "k_ik<7, 0, 13, 9265531810339127745ull>"
K_RNE_PANDA = "k_rne<7, true, true, 16140979858257510057ull>"


def _committed(root, out):
    """`frac_rocprof_committed` for the secondary legs, from the rocprofv3 --kernel-trace run of THIS command committed under profiles/
    (rNN_*_secondary_kernel_stats.csv: scripts/visit.sh stage `profsec` -> scripts/secondary_stats.py, durations per kernel AND grid size -- config 4
    launches the same k_rne at 1e7 triples and at its 1.25e6 share): the event figures above, cross-checked by the profiler on another lease."""
    import csv
    import glob
    import os
    files = sorted(glob.glob(os.path.join(root, "profiles", "r[0-9][0-9]_*_secondary_kernel_stats.csv")))
    if not files:
        return
    rows = list(csv.DictReader(open(files[-1])))
    src = "profiles/" + os.path.basename(files[-1])

    def find(sub, grid=None):
        hits = [r for r in rows if sub.replace(" ", "") in (r.get("Name") or "").replace(" ", "") and (grid is None or int(r["GridSize"]) == grid)]
        return max(hits, key=lambda r: float(r["TotalDurationNs"])) if hits else None

    def note(row):
        return {"file": src, "kernel": row["Name"][:120], "grid": int(row["GridSize"]), "avg_ns": float(row["AverageNs"]), "calls": int(row["Calls"])}

    def hbm(leg, row):
        if row is None or not isinstance(out.get(leg), dict) or "roofline" not in out[leg]:
            return
        rf = out[leg]["roofline"]
        rf["frac_rocprof_committed"] = rf["algorithmic_bytes_per_launch"] / (float(row["AverageNs"]) * 1e-9) / 1e9 / HBM_PEAK_GBS
        rf["rocprof_committed"] = note(row)
    for leg in ("rne_config4_1e7", "rne_config4_shard"):
        if isinstance(out.get(leg), dict) and "n" in out[leg]:
            hbm(leg, find(K_RNE_PANDA, ((out[leg]["n"] + 63) // 64) * 64))
    ik = find(K_IK_PANDA)
    leg = out.get("ik_config3")
    if ik is not None and isinstance(leg, dict) and "roofline" in leg:
        its = leg["mean_iterations"] * leg["n"]
        merge = find("k_ik_merge_flat")
        leg["roofline"]["frac_rocprof_committed"] = its * IK_FLOPS_PER_ITERATION / (float(ik["AverageNs"]) * 1e-9) / 1e12 / FP64_VALU_PEAK_TFLOPS
        leg["roofline"]["rocprof_committed"] = dict(note(ik), merge_kernel=None if merge is None else note(merge),
                                                    note="the scheduler kernel alone; a call is this + two fills + k_ik_merge_flat")
    f0, f1 = find("k_fleet<0"), find("k_fleet<1")
    leg = out.get("fleet_config5")
    if f0 is not None and f1 is not None and isinstance(leg, dict) and "roofline" in leg:
        # one call = one launch of each class: the leg's bytes over the sum of the two average durations
        # (rows of the 17-chain form: the largest grids of each kernel)
        t = (float(f0["AverageNs"]) + float(f1["AverageNs"])) * 1e-9
        leg["roofline"]["frac_rocprof_committed"] = leg["roofline"]["algorithmic_bytes_per_launch"] / t / 1e9 / HBM_PEAK_GBS
        leg["roofline"]["rocprof_committed"] = {"k_fleet<0>": note(f0), "k_fleet<1>": note(f1)}


def secondary(rtbhip):
    """{"ik_config3", "rne_config4_1e7", "rne_config4_shard", "fleet_config5", "seconds"}; a leg that cannot run reports {"error": ...}
    (a PARITY failure is not such an error: it ends the bench)."""
    t_all = time.perf_counter()
    out = {}
    for key, fn in (("ik_config3", ik_config3), ("rne_config4", rne_config4), ("fleet_config5", fleet_config5), ("structure", structure_legs)):
        t0 = time.perf_counter()
        try:
            r = fn(rtbhip)
        except SystemExit:
            raise
        except Exception as e:                                  # e.g. out of memory on a shared GPU: say so, keep the headline
            r = {"error": repr(e)[:300]}
        if key == "rne_config4":
            if isinstance(r, tuple):
                out["rne_config4_1e7"], out["rne_config4_shard"] = r
                out["rne_config4_1e7"]["seconds"] = time.perf_counter() - t0
            else:
                out["rne_config4_1e7"] = out["rne_config4_shard"] = r
        else:
            out[key] = r
            if isinstance(r, dict):
                r["seconds"] = time.perf_counter() - t0
    import os
    try:
        _committed(os.path.dirname(os.path.abspath(__file__)), out)
    except Exception as e:                                       # a malformed profile file must not cost the run its line
        out["_profiles"] = {"error": repr(e)[:200]}
    out["seconds"] = time.perf_counter() - t_all
    return out

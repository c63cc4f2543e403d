#!/usr/bin/env python3
"""bench.py -- headline benchmark: Franka Panda 7-DOF fused fkine+jacob0, configurations / second.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 launched by torch.distributed.run,
one rank per GPU).  A "step" is one pass of the hot path (ONE fused kernel launch) over one batch
of synthetic input that is already resident in HBM: BASELINE.json configs[1] -- ETS Panda, 22 ETs,
q ~ U(-pi,pi)^7, N = 1e6 per GPU, fp64.  W untimed warm-up steps, then exactly K timed steps between
barrier+synchronize pairs; the MAX over ranks is the step time; rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     achieved = 520 B/config x N / (average duration of the dominant kernel = the device-side duration
               of the K timed launches, ONE HIP-event pair on the launch stream around the timed loop, / K)
               against the 8 TB/s HBM3E peak; traffic = HBM bytes per launch from the committed rocprofv3 --pmc
               passes of this command (profiles/r06_pmc.json; `traffic_source` says so -- it is not measured
               by this run), else null.
  host_path    (N=1 only, never `value`) the same 1e6 configurations handed over as NumPy arrays through the
               RTBHIP_MEM_HOST boundary: PCIe-inclusive configurations/s of the pinned, chunked, double-buffered
               H2D -> kernel -> D2H pipeline inside the library.
  cpu_baseline the reference's OWN native code (oracle/_ref, built unmodified from the reference
               sources) timed on this box's host cores on the same q array: one ETS_fkine call over
               the whole array + the Python per-row ETS_jacob0 loop a reference user needs today
               (kind "reference"); falls back to the plain-C restatement (kind "port").
Multi-GPU: the batch dimension is embarrassingly parallel -> each rank owns N rows (weak scaling),
no collective on the data path; the single output gather (RCCL all_gather over xGMI) is timed
separately and reported as gather_ms, never inside `value`.  `--gpus N` with no launcher (WORLD_SIZE unset)
re-executes this script under `python -m torch.distributed.run --nproc-per-node N`; under a launcher the world
must equal --gpus.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]

from benchlib import protect_stdout, HBM_PEAK_GBS, Ranks, spawn_ranks_if_needed, bench_argv, per_launch_min_ms, ensure_library, pmc_traffic, rocprof_committed, rocprof_committed_all  # noqa: E402

BYTES_PER_CONFIG = 56 + 128 + 336  # q read + T written + J0 written
KERNEL = {"packed": "k_kin_reg<7,true,true,true> (rtbhip_fkine_jacob_packed: [T | J0] rows of one (N,58) array)",
          "two": "k_kin_reg<7,true,true,false> (rtbhip_fkine_jacob: T (N,4,4) and J0 (N,6,7))"}
KERNEL_SUBSTR = {"packed": "k_kin_reg<7, true, true, true>", "two": "k_kin_reg<7, true, true, false>"}


def cpu_baseline_all_cores(sample, timeout_s=90.0):
    """the same reference path on every host core.  Runs in a separate, time-limited process
    (oracle/cpu_pool_bench.py: a fork pool with one reference chain per worker) so that nothing it does -- or
    fails to do -- can stall this one; any failure is reported instead of a number."""
    import subprocess
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "q.npy")
        np.save(path, sample)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_pool_bench.py"), path],
                           capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
    if r.returncode != 0:
        raise RuntimeError("cpu_pool_bench.py rc=%d: %s" % (r.returncode, r.stderr.strip()[-300:]))
    return json.loads(r.stdout.strip().splitlines()[-1])


def cpu_baseline(q_host, T_gpu, J_gpu, max_seconds=25.0):
    """Reference CPU path on this box (1 core: the extension holds the GIL and has no threads).
    The same pass doubles as the parity guard of the timed configuration: the GPU's T/J rows of the
    sample are compared with what the CPU leg just produced."""
    import numpy as np
    from oracle import chains
    ch = chains.panda_ets()
    try:
        from oracle import ref_harness
        if not ref_harness.available():
            raise ImportError("oracle/_ref not built")
        ref = ref_harness.RefETS(ch)
        kind = "reference"
        fk = lambda a: ref.fkine(a)
        jc = lambda a: ref.jacob0_batch(a)
    except Exception:
        from oracle import oracle
        kind = "port"
        fk = lambda a: oracle.fkine(ch, a)
        jc = lambda a: oracle.jacob0(ch, a)
    n = min(len(q_host), 200000)
    sample = np.ascontiguousarray(q_host[:n])
    fk(sample[:1000]); jc(sample[:1000])  # warm-up
    reps, t_used, best = 0, 0.0, None
    while reps < 5 and t_used < max_seconds:
        t0 = time.perf_counter()
        Tc = fk(sample)
        Jc = jc(sample)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        t_used += dt
        reps += 1
    err = max(float(np.abs(T_gpu[:n].cpu().numpy() - Tc).max()), float(np.abs(J_gpu[:n].cpu().numpy() - Jc).max()))
    if not err <= 1e-10:
        raise SystemExit("bench: parity check vs the CPU %s failed, max |err| = %g" % (kind, err))
    out = {"value": n / best, "unit": "configurations/s", "cores": 1, "kind": kind, "max_abs_err_gpu_vs_cpu": err,
           "sample": "first %d of the %d bench configurations, best of %d passes; ETS_fkine over the array "
                     "+ per-row ETS_jacob0 loop (the reference has no batched Jacobian)" % (n, len(q_host), reps)}
    if kind == "reference" and (os.cpu_count() or 1) > 1:
        try:
            out["all_cores"] = cpu_baseline_all_cores(q_host)      # every configuration of the step: enough rows per core
        except Exception as e:                    # the single-core figure above is the contract; this one is a bonus
            out["all_cores"] = {"error": repr(e)}
    return out


def host_path(ets, q_host, T_dev, J_dev, reps=3):
    """PCIe-inclusive rate of the host-pointer boundary (what `panda.fkine(numpy_q)` uses): NumPy in, NumPy out."""
    import numpy as np
    N = len(q_host)
    ets.fkine_jacob0(q_host[:4096])                   # staging buffers / streams come up once
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = ets.fkine_jacob0(q_host)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    err = max(float(np.abs(out[0] - T_dev.cpu().numpy()).max()), float(np.abs(out[1] - J_dev.cpu().numpy()).max()))
    if err != 0.0:
        raise SystemExit("bench: host-pointer path differs from the device-pointer path, max |err| = %g" % err)
    return {"value": N / best, "unit": "configurations/s", "ms_per_call": best * 1e3, "pcie_GBs": BYTES_PER_CONFIG * N / best / 1e9,
            "what": "rtbhip_fkine_jacob(RTBHIP_MEM_HOST) on %d pageable NumPy configurations in, T and J0 NumPy arrays out; "
                    "best of %d calls; bit-equal to the device-pointer results" % (N, reps)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1000000, help="configurations per GPU per step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and host_path legs")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` object (BASELINE configs[2], [3], [4] with their in-run parity)")
    ap.add_argument("--tune", action="append", default=[], help="key=value for rtbhip_tune (A/B runs)")
    ap.add_argument("--layout", choices=("packed", "two"), default="two",
                    help="output layout of the timed step: 'packed' = rtbhip_fkine_jacob_packed, one (N,58) array of [T | J0] rows (a single write "
                         "stream; the T||J gather message of SURVEY 8e); 'two' = rtbhip_fkine_jacob, T (N,4,4) and J0 (N,6,7) as two arrays.  Same "
                         "arithmetic, same 520 algorithmic bytes per configuration, bit-identical values (the other layout is timed beside it)")
    ap.add_argument("--gather", action="store_true", help="build the process group and time the output gather even with one rank "
                                                          "(a world-size-1 RCCL group: rehearses the collective on a single-GPU box)")
    argv = bench_argv()
    args = ap.parse_args(argv)
    if args.gather:
        os.environ["RTBHIP_BENCH_FORCE_GROUP"] = "1"
    spawn_ranks_if_needed(args.gpus, os.path.abspath(__file__), argv)
    protect_stdout()

    import numpy as np
    import torch
    ensure_library(ROOT)
    import rtbhip

    rk = Ranks()
    rank, world, dev = rk.rank, rk.world, rk.dev
    for kv in args.tune:
        k, v = kv.split("=")
        rtbhip.tune(k, int(v))

    N = args.n
    robot = rtbhip.models.Panda()
    ets = robot.ets()
    rng = np.random.default_rng(rank)  # seed 0 on rank 0 = BASELINE config 2
    q_host = rng.uniform(-np.pi, np.pi, (N, 7))
    q = torch.from_numpy(q_host).to(dev)
    packed = args.layout == "packed"
    lib = rtbhip.lib()
    import ctypes as C
    h = ets._handle()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    qp = C.c_void_p(q.data_ptr())

    def make_outputs(layout):
        """(buffers, step function over them): 'packed' one (N,58) array, 'two' the (N,4,4) + (N,6,7) pair"""
        if layout == "packed":
            TJ = torch.empty((N, 58), dtype=torch.float64, device=dev)
            tjp = C.c_void_p(TJ.data_ptr())

            def go():
                rc = lib.rtbhip_fkine_jacob_packed(h, qp, N, None, None, 0, tjp, 1, stream)
                if rc != 0:
                    raise RuntimeError(lib.rtbhip_last_error().decode())
            return (TJ,), go
        T_ = torch.empty((N, 4, 4), dtype=torch.float64, device=dev)
        J_ = torch.empty((N, 6, 7), dtype=torch.float64, device=dev)
        tp, jp = C.c_void_p(T_.data_ptr()), C.c_void_p(J_.data_ptr())

        def go():
            rc = lib.rtbhip_fkine_jacob(h, qp, N, None, None, 0, tp, jp, 1, stream)
            if rc != 0:
                raise RuntimeError(lib.rtbhip_last_error().decode())
        return (T_, J_), go

    outs, step = make_outputs(args.layout)
    if packed:
        TJ = outs[0]
        T, J = TJ[:, :16].unflatten(1, (4, 4)), TJ[:, 16:].unflatten(1, (6, 7))        # strided views of the rows
    else:
        T, J = outs
        TJ = None

    elapsed, kern_avg_ms = rk.timed_steps(step, args.steps, args.warmup)
    kern_min_ms = per_launch_min_ms(step, min(args.steps, 50))
    rotating, layouts = None, None
    if rank == 0 and world == 1 and not args.no_secondary:      # (--no-secondary: the profiling runs, whose kernel averages are of the timed loop alone)
        # The timed loop above rewrites ONE output buffer set, as a control loop with preallocated outputs does.  Beside it, for BOTH layouts:
        # the sustained rate on one buffer set, and on THREE sets used in rotation -- every launch writes memory the previous two did not touch,
        # the rate at which results stream into memory that has to take them.  The two-array form's rate depends on where the allocator put
        # its two arrays relative to one another (10-14 %, profiles/r04_headline_stores.txt) and, on one set, on the pose array fitting the
        # memory-side cache; the packed form is a single stream and has neither dependence.  Reported, never `value`.
        from benchlib import sustained_ms
        layouts = {}
        for name in ("packed", "two"):
            sets = [(outs, step)] if name == args.layout else [make_outputs(name)]
            sets += [make_outputs(name) for _ in range(2)]
            state = {"k": 0}

            def rot():
                sets[state["k"] % 3][1]()
                state["k"] += 1
            rot()
            rms, _, _ = sustained_ms(rot)
            sets[0][1]()
            sms, _, _ = sustained_ms(sets[0][1])
            frac = lambda ms: BYTES_PER_CONFIG * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            layouts[name] = {"one_set_sustained_ms": sms, "one_set_sustained_frac": frac(sms), "rotating3_ms": rms, "rotating3_frac": frac(rms)}
            del sets
        layouts["what"] = ("sustained timing (>= 30 ms of back-to-back launches after >= 30 ms of warm-up) of rtbhip_fkine_jacob_packed ('packed': one "
                           "(N,58) array) and rtbhip_fkine_jacob ('two': T and J0 arrays) on one output set and on three sets in rotation; same values bit for bit")
        cur = layouts[args.layout]
        rotating = {"output_pairs": 3, "kernel_avg_ms": cur["rotating3_ms"], "frac": cur["rotating3_frac"],
                    "one_pair_sustained_ms": cur["one_set_sustained_ms"], "one_pair_sustained_frac": cur["one_set_sustained_frac"],
                    "what": "the timed layout (%s) under sustained timing: three output sets in rotation / one set" % args.layout}
        torch.cuda.empty_cache()
    probe = None
    if rank == 0 and world == 1:
        # what the memory system of THIS box delivers for the same traffic (56 B read + 464 B written per configuration) through a plain
        # streaming kernel, measured the same way (events around 20 launches): the spread between boxes is 15-20 %, the data-sheet peak is not
        dst = torch.empty((N, 58), dtype=torch.float64, device=dev)
        sp, dp = C.c_void_p(q.data_ptr()), C.c_void_p(dst.data_ptr())

        def probe_step():
            rc = lib.rtbhip_stream_probe(sp, N * 7 & ~1, dp, N * 58 & ~1, stream)
            if rc != 0:
                raise RuntimeError(lib.rtbhip_last_error().decode())
        for _ in range(3):
            probe_step()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            probe_step()
        e1.record()
        torch.cuda.synchronize()
        pms = e0.elapsed_time(e1) / 20
        moved = ((N * 7) // 512 + (N * 58) // 512) * 4096           # whole 4 KiB pages (the kernel leaves the tails alone)
        probe = {"kernel": "k_stream_probe: single-wave workgroups, one aligned 4 KiB page of 16-byte non-temporal stores each (the best pattern of "
                           "profiles/r01_k_write_probe.txt = the rate of hipMemsetAsync), every 8th also loads a page; 56 B read + 464 B written per "
                           "configuration, no arithmetic",
                 "ms": pms, "GBs": moved / (pms * 1e-3) / 1e9}
        del dst
    line = None
    if rank == 0:                                   # the line is complete BEFORE the exchange below is tried (see the watchdog)
        achieved = BYTES_PER_CONFIG * N / (kern_avg_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        for name in ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):       # the latest committed PMC passes of this command
            traffic, traffic_source = pmc_traffic(ROOT, name)
            if traffic is not None:
                break
        line = {
            "metric": "configurations/sec (Panda 7-DOF fkine+jacob0)",
            "value": world * N * args.steps / elapsed,
            "unit": "configurations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # t0 (after the opening barrier) -> each rank's OWN synchronize, MAX over ranks: no collective inside (benchlib.Ranks.timed_steps).
            # Beside it the same clock read after the closing barrier (for one rank: the same number).
            "ms_per_step_with_barrier": rk.last_elapsed_with_barrier / args.steps * 1e3,
            "timed_region": "opening barrier+sync | t0 | K launches | own torch.cuda.synchronize | t1 | closing barrier; value = units / MAX over ranks (t1 - t0)",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: ETS Panda (22 ETs, 7 joints) fused fkine+jacob0, "
                                   "q~U(-pi,pi)^7 seed 0, N=%d per GPU, fp64, device-resident" % N,
                       "configs_per_gpu": N, "sharding": "rows/%d, no data-path collective" % world,
                       "backend": rk.backend if rk.dist is not None else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": KERNEL[args.layout], "layout": args.layout, "kernel_avg_ms": kern_avg_ms, "per_launch_event_min_ms": kern_min_ms,
                         "kernel_avg_source": "one HIP-event pair on the launch stream around the K timed launches / K "
                                              "(per_launch_event_min_ms: smallest of per-launch event pairs, each of which adds a few microseconds "
                                              "-- it can exceed the loop average)",
                         "algorithmic_bytes_per_launch": BYTES_PER_CONFIG * N},
        }
        committed = rocprof_committed(ROOT, KERNEL_SUBSTR[args.layout])
        if committed is not None:
            # the rocprofv3 --kernel-trace --stats figure of the SAME command committed under profiles/ (another lease, possibly another box:
            # boxes of this pool differ by 10-15 % on this kernel) next to what this run measured with events
            line["roofline"]["frac_rocprof_committed"] = BYTES_PER_CONFIG * N / (committed["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBS
            line["roofline"]["rocprof_committed"] = committed
            line["roofline"]["rocprof_committed_leases"] = [dict(x, frac=BYTES_PER_CONFIG * N / (x["avg_ns"] * 1e-9) / 1e9 / HBM_PEAK_GBS)
                                                            for x in rocprof_committed_all(ROOT, KERNEL_SUBSTR[args.layout])]      # the spread between leases (boxes)
        if rotating is not None:
            line["roofline"]["rotating_outputs"] = rotating
            line["roofline"]["layouts"] = layouts
        if probe is not None:
            # context, not a ceiling: a plain streaming kernel with the same read / write mix on THIS box (the fused kernel has beaten it)
            line["roofline"]["stream_probe"] = probe
        if rk.shared:
            line["config"]["devices_shared"] = True   # gloo test hook: more ranks than GPUs, NOT a scaling measurement
    # every rank's device identity and its own step / kernel times (the line answers "did N ranks drive N GPUs?" by itself)
    # Watchdog: the exchange below is the one part of a multi-rank run that no single-GPU box can rehearse at N > 1 (RCCL over xGMI).  If it has
    # not come back after RTBHIP_BENCH_GATHER_TIMEOUT seconds (default 180) rank 0 prints the line it already holds -- `value`, the roofline --
    # with the reason in place of the gather figures, and every rank leaves: a stuck collective must not cost the run its measurement.
    watchdog = None
    if rk.dist is not None and world > 1:
        import threading

        def give_up():
            if rank == 0:
                line["gather"] = "NOT MEASURED: the rank identities / output gather did not complete within %s s (watchdog)" % os.environ.get("RTBHIP_BENCH_GATHER_TIMEOUT", "180")
                print(json.dumps(line), flush=True)
            sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(float(os.environ.get("RTBHIP_BENCH_GATHER_TIMEOUT", "180")), give_up)
        watchdog.daemon = True
        watchdog.start()
    if os.environ.get("RTBHIP_BENCH_TEST_STALL") and world > 1:       # TEST HOOK (tests/test_dist_gpu.py; never set by the driver): the exchange hangs
        time.sleep(float(os.environ["RTBHIP_BENCH_TEST_STALL"]))
    per_rank = rk.identities(ms_per_step_own=rk.last_own_elapsed / args.steps * 1e3, kernel_avg_ms=kern_avg_ms) if rk.dist is not None else None
    # the one exchange of the path, outside the timed region: the ranks' T||J rows to rank 0 (and, beside it, to every rank).  With the packed
    # layout the rows ARE the kernel's output; with two arrays one more packed launch makes them (no torch.cat copy either way).
    gather_ms, gather_mem = None, None
    if rk.dist is not None:
        if packed:
            rows = TJ
        else:
            (rows,), go = make_outputs("packed")
            go()
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        gather_ms = rk.gather_ms(rows)                    # allocates its N_total x 58 receive buffers inside, releases them on return
        gather_mem = dict(rk.last_gather, device_bytes_before=before, device_bytes_after=torch.cuda.memory_allocated(),
                          where="after the timed region of `value`, released before cpu_baseline / secondary")
        if not packed:
            del rows
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0:
        if per_rank is not None:
            line["ranks"] = per_rank
            line["world"] = {"launcher": world, "process_group": rk.dist.get_world_size(), "backend": rk.backend,
                             "rccl_world": (gather_mem or {}).get("rccl_world"), "distinct_devices": len({r["uuid"] + r["pci_bus_id"] for r in per_rank})}
        if gather_ms is not None:
            line["gather_ms"] = gather_ms                       # gather-to-root (rank 0), the default form of the exchange
            line["all_gather_ms"] = gather_mem.get("all_gather_ms")
            line["gather_buffer"] = gather_mem
            line["gather"] = "gather of the ranks' (N,58) f64 T||J rows to rank 0 (gather_ms) and to every rank (all_gather_ms), %s" % (
                ("RCCL through rtbhip_shard_gather, a world-size-1 communicator on one GPU (--gather: rehearsal of the collective, no xGMI traffic)" if world == 1
                 else "RCCL over xGMI through rtbhip_shard_gather")
                if rk.backend == "nccl" else "gloo through host memory (test hook)")
        if not args.no_cpu and world == 1:  # reported on rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(q_host, T, J)
            line["host_path"] = host_path(ets, q_host, T, J)
        if not args.no_secondary and world == 1 and not rk.forced:
            # BASELINE configs[2], [3], [4] -- IK over 1e5 targets, RNE over 1e7 triples (and the 1.25e6 share), the 16-arm fleet -- each with
            # its HIP-event kernel time, an in-run parity figure against the reference's compiled code on a sample, and its roofline;
            # after the timed region, never inside `value` (benchsecondary.py)
            del T, J, q
            torch.cuda.empty_cache()
            import benchsecondary
            line["secondary"] = benchsecondary.secondary(rtbhip)
        print(json.dumps(line), flush=True)
    rk.finish()


if __name__ == "__main__":
    main()

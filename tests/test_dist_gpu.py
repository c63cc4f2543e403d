"""-m gpu: the N>1 path with REAL kernels -- two ranks (one process each, gloo rendezvous on 127.0.0.1) share the
box's one GPU; each launches rtbhip_fkine_jacob and rtbhip_rne on its own row block (rtbhip_shard_range through
ShardedBatch), the shards are gathered with the path's one collective, and the full result is compared with the
CPU oracle -- including N not divisible by the world size and a world larger than N.  On an 8-GPU node the same
code runs with backend "nccl" (RCCL over xGMI), one GPU per rank; only the backend string differs.
Also: `bench.py --gpus 2` spawns two ranks by itself and says so on its JSON line."""
import json
import os
import socket
import subprocess
import time
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, to_all, outq):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
    import torch
    import torch.distributed as dist
    import rtbhip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank derives the SAME global inputs from the seed, then keeps only its rows
        rng = np.random.default_rng(99)
        qg = rng.uniform(-np.pi, np.pi, (N, 7))
        arm = rtbhip.models.DH.Panda()
        qa = rng.uniform(arm.qlim[0], arm.qlim[1], (N, 7))
        qd, qdd = rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
        sb = rtbhip.ShardedBatch(N)
        ets = rtbhip.models.Panda().ets()
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(sb.local(a))).cuda()
        T, J = ets.fkine_jacob0(dev(qg))
        tau = arm.rne(dev(qa), dev(qd), dev(qdd))
        # a one-row block comes back as ONE configuration ((4,4), (6,7), (7,)): the reference's shape rule for (1,n) input
        T, J, tau = T.reshape(sb.count, 4, 4), J.reshape(sb.count, 6, 7), tau.reshape(sb.count, 7)
        assert T.is_cuda
        TJ = torch.cat([T.reshape(sb.count, 16), J.reshape(sb.count, 42)], dim=1)
        full = sb.gather(TJ, to_all=to_all)
        ftau = sb.gather(tau, to_all=to_all)
        if to_all or rank == 0:
            outq.put((rank, sb.begin, sb.count, full.cpu().numpy(), ftau.cpu().numpy()))
        else:
            assert full is None and ftau is None
            outq.put((rank, sb.begin, sb.count, None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,world,to_all", [(4097, 2, True), (1000, 2, False), (130, 3, True), (2, 3, True)])
def test_two_ranks_real_kernels_gathered_equal_oracle(N, world, to_all):
    from oracle import oracle, chains
    ctx = mp.get_context("spawn")
    outq = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, to_all, outq)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([outq.get(timeout=300) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # the row blocks tile [0, N) in rank order
    assert res[0][1] == 0 and all(res[i][1] + res[i][2] == res[i + 1][1] for i in range(world - 1))
    assert res[-1][1] + res[-1][2] == N
    rng = np.random.default_rng(99)
    qg = rng.uniform(-np.pi, np.pi, (N, 7))
    tab = chains.panda_dh()
    qa = rng.uniform(tab.qlim[:, 0], tab.qlim[:, 1], (N, 7))
    qd, qdd = rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
    ch = chains.panda_ets()
    Tref, Jref = oracle.fkine(ch, qg), oracle.jacob0(ch, qg)
    tref = oracle.rne_dh(tab.L24(), 1, qa, qd, qdd, -tab.gravity)
    holders = [r for r in res if r[3] is not None]
    assert len(holders) == (world if to_all else 1)
    for _, _, _, full, ftau in holders:
        assert full.shape == (N, 58) and ftau.shape == (N, 7)
        assert np.abs(full[:, :16].reshape(N, 4, 4) - Tref).max() <= 1e-10
        assert np.abs(full[:, 16:].reshape(N, 6, 7) - Jref).max() <= 1e-10
        assert np.abs(ftau - tref).max() / max(1.0, np.abs(tref).max()) <= 1e-9


def _nccl_world1_worker(port, outq):
    """ONE rank, backend "nccl" (= RCCL): communicator set-up, all_reduce, all_gather_into_tensor and gather on DEVICE buffers
    that rtbhip_fkine_jacob / rtbhip_rne just filled -- the code path an 8-GPU node runs, on the one GPU of this box."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
    import torch
    import torch.distributed as dist
    import rtbhip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        N = 4097
        rng = np.random.default_rng(99)
        qg = rng.uniform(-np.pi, np.pi, (N, 7))
        sb = rtbhip.ShardedBatch(N)
        assert (sb.rank, sb.world, sb.begin, sb.count) == (0, 1, 0, N)
        ets = rtbhip.models.Panda().ets()
        T, J = ets.fkine_jacob0(torch.from_numpy(qg).cuda())
        TJ = torch.cat([T.reshape(N, 16), J.reshape(N, 42)], dim=1)
        a = sb.gather(TJ, to_all=True, collective="always")            # all_gather_into_tensor on the device buffer
        b = sb.gather(TJ, to_all=False, dst=0, collective="always")    # dist.gather to rank 0
        assert a.is_cuda and b.is_cuda and a.data_ptr() != TJ.data_ptr() and b.data_ptr() != TJ.data_ptr()
        t = torch.tensor([3.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        loaded = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
        outq.put((a.cpu().numpy(), b.cpu().numpy(), TJ.cpu().numpy(), float(t.item()), sorted(set(loaded))))
    finally:
        dist.destroy_process_group()


def test_nccl_world1_gathers_device_buffers_through_rccl():
    from oracle import oracle, chains
    ctx = mp.get_context("spawn")
    outq = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), outq))
    p.start()
    a, b, tj, red, loaded = outq.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert loaded, "librccl was not mapped: the collective did not go through RCCL"
    assert np.array_equal(a, tj) and np.array_equal(b, tj) and red == 3.5
    qg = np.random.default_rng(99).uniform(-np.pi, np.pi, (4097, 7))
    ch = chains.panda_ets()
    assert np.abs(a[:, :16].reshape(-1, 4, 4) - oracle.fkine(ch, qg)).max() <= 1e-10
    assert np.abs(a[:, 16:].reshape(-1, 6, 7) - oracle.jacob0(ch, qg)).max() <= 1e-10


def test_bench_gather_flag_runs_the_rccl_gather_with_one_rank():
    """`torchrun --nproc-per-node 1 bench.py --gpus 1 --gather`: backend nccl, world 1, gather_ms on the line."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RTBHIP_BENCH_BACKEND", "RTBHIP_BENCH_FORCE_GROUP"):
        env.pop(k, None)
    # the script's flags travel in RTBHIP_BENCH_ARGV, as in benchlib.spawn_ranks_if_needed (torch.distributed.run's parser claims
    # abbreviations of its own options even after the script name)
    env["RTBHIP_BENCH_ARGV"] = json.dumps(["--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu", "--gather"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["gather_ms"] > 0 and "RCCL" in d["gather"]
    # the receive buffer of the gather (world x N x 58 doubles) lives only inside Ranks.gather_ms: after the timed region of `value`
    # (test_dist_gloo.py checks the call order), gone again when the call returns -- the device holds what it held before
    gb = d["gather_buffer"]
    assert gb["buffer_bytes"] == 1 * 1000000 * 58 * 8 and gb["device_bytes_after"] == gb["device_bytes_before"]
    assert "secondary" not in d                        # the forced one-rank group is a rehearsal of the collective: no secondary legs
    # both forms of the exchange, through the C ABI (rtbhip_shard_gather -> RCCL), and the line's own account of who ran where
    assert d["all_gather_ms"] > 0 and "rtbhip_shard_gather" in gb["via"] and gb["rccl_world"] == 1 and gb["padded"] is False
    assert d["world"] == {"launcher": 1, "process_group": 1, "backend": "nccl", "rccl_world": 1, "distinct_devices": 1}
    (r0,) = d["ranks"]
    assert r0["rank"] == 0 and len(r0["uuid"]) == 32 and ":" in r0["pci_bus_id"] and r0["kernel_avg_ms"] > 0 and r0["ms_per_step_own"] > 0


def _run_bench(script, extra):
    env = dict(os.environ, RTBHIP_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "2"] + extra, capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_gpus_2_spawns_two_ranks():
    lines = _run_bench("bench.py", ["--steps", "5", "--warmup", "2", "--n", "200000", "--no-cpu"])
    assert len(lines) == 1
    line = lines[0]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["gather_ms"] > 0
    assert line["config"]["configs_per_gpu"] == 200000
    assert line["roofline"]["kernel_avg_ms"] <= line["ms_per_step"] * 1.001     # device time of the loop <= host time of the loop
    assert line["value"] == pytest.approx(2 * 200000 * 5 / (line["ms_per_step"] * 5e-3), rel=1e-9)


def test_bench_extra_gpus_2_shards_config4_config3_and_config5():
    lines = _run_bench("bench_extra.py", ["--what", "rne,ik,fleet", "--steps", "4", "--n-rne", "200001", "--n-ik", "20001", "--n-fleet", "20001", "--no-cpu"])
    assert len(lines) == 3
    rne, ik, fleet = lines
    assert rne["n_gpus"] == 2 and rne["n"] == 200001 and rne["rows_rank0"] == 100001 and rne["scaling"] == "strong" and rne["gather_ms"] > 0
    assert ik["n_gpus"] == 2 and ik["n"] == 20001 and ik["rows_rank0"] == 10001 and 0.97 < ik["success_rate"] <= 1.0 and ik["roofline"]["bound"] == "fp64-valu"
    assert fleet["n_gpus"] == 2 and fleet["scaling"] == "strong" and len(fleet["arms"]) == 16


def test_bench_eight_ranks_sharing_the_gpu():
    """The driver's 8-GPU run in miniature: `bench.py --gpus 8` and the sharded legs of `bench_extra.py --gpus 8` with eight ranks SHARING this
    box's GPU (the gloo test hook: `devices_shared` is printed, this is no scaling figure).  What it checks is the eight-rank arithmetic:
    row blocks of a batch that eight does not divide, the gather of eight uneven shards, max-over-ranks timing, one JSON line from rank 0."""
    def run(script, extra):
        env = dict(os.environ, RTBHIP_BENCH_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--gpus", "8"] + extra, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        return [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    lines = run("bench.py", ["--steps", "4", "--warmup", "1", "--n", "100003", "--no-cpu"])
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["devices_shared"] is True and d["gather_ms"] > 0
    assert d["gather_buffer"]["world"] == 8 and d["gather_buffer"]["buffer_bytes"] == 8 * 100003 * 58 * 8
    # the self-verifying part of the line: eight ranks, each with its device identity and its own times; one physical GPU here (and said so)
    assert d["world"]["launcher"] == 8 and d["world"]["process_group"] == 8 and d["world"]["backend"] == "gloo" and d["world"]["distinct_devices"] == 1
    assert [r["rank"] for r in d["ranks"]] == list(range(8)) and d["all_gather_ms"] > 0
    assert all(len(r["uuid"]) == 32 and r["kernel_avg_ms"] > 0 and r["ms_per_step_own"] > 0 for r in d["ranks"])
    assert d["value"] == pytest.approx(8 * 100003 * 4 / (d["ms_per_step"] * 4e-3), rel=1e-9) and "secondary" not in d
    rne, ik, fleet = run("bench_extra.py", ["--what", "rne,ik,fleet", "--steps", "3", "--n-rne", "400003", "--n-ik", "40003", "--n-fleet", "20003", "--no-cpu"])
    assert rne["n_gpus"] == 8 and rne["n"] == 400003 and rne["rows_rank0"] == 50001 and rne["gather_ms"] > 0
    assert ik["n_gpus"] == 8 and ik["n"] == 40003 and ik["rows_rank0"] == 5001 and 0.97 < ik["success_rate"] <= 1.0
    assert fleet["n_gpus"] == 8 and len(fleet["arms"]) == 16


def test_bench_watchdog_keeps_the_measurement_when_the_exchange_hangs():
    """The output gather is the one part of `bench.py --gpus N` a single-GPU box cannot rehearse at N > 1 over RCCL.  If it never comes back, rank 0
    prints the line it already holds -- value, roofline -- with the reason in place of the gather figures, and all ranks leave (exit 0).
    Here: two ranks on the gloo hook, the exchange stalled by the test hook for longer than a 2 s watchdog."""
    env = dict(os.environ, RTBHIP_BENCH_BACKEND="gloo", RTBHIP_BENCH_TEST_STALL="30", RTBHIP_BENCH_GATHER_TIMEOUT="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--n", "100000", "--no-cpu"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert time.time() - t0 < 120
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = lines[0]
    assert d["n_gpus"] == 2 and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1 and "gather_ms" not in d
    assert d["gather"].startswith("NOT MEASURED") and "watchdog" in d["gather"]


def test_bench_gpus_8_on_a_smaller_box_fails_with_one_clear_line():
    """`python bench.py --gpus 8` where fewer than 8 GPUs exist: every rank refuses with the same one-line reason (no silent sharing, no hang):
    the RCCL path wants one device per rank."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("this box has 8 GPUs: the run would simply work")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RTBHIP_BENCH_BACKEND", "RTBHIP_BENCH_FORCE_GROUP"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]             # no JSON line: nothing was measured
    assert "8 ranks but %d GPUs" % torch.cuda.device_count() in r.stderr


def test_bench_single_gpu_line_carries_the_contract_objects():
    """`python bench.py` as the driver runs it (N = 1, default workload): ONE JSON line with the contract's keys, the roofline
    object priced on 520 B per configuration, the reference-built CPU baseline, and the host-pointer rate as its own object."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RTBHIP_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3"], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                                     # exactly one line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["dtype"] == "f64" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(1e6 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and 0.3 < rf["frac"] < 1.0
    assert rf["achieved"] == pytest.approx(520e6 / (rf["kernel_avg_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert rf["kernel_avg_ms"] <= d["ms_per_step"] * 1.001
    assert rf["traffic"] is None or (0.95 * 520e6 < rf["traffic"] < 1.1 * 520e6 and "profiles/" in rf["traffic_source"])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 1 and cb["value"] > 1e5 and cb["max_abs_err_gpu_vs_cpu"] < 1e-10 and "sample" in cb
    hp = d["host_path"]
    assert hp["value"] > 1e7 and hp["value"] < d["value"] and hp["unit"] == "configurations/s"
    # BASELINE configs[2], [3], [4] on the same line, each with its in-run parity figure against the reference's compiled code
    sec = d["secondary"]
    assert sec["seconds"] < 60
    ik, rne, shard, fleet = sec["ik_config3"], sec["rne_config4_1e7"], sec["rne_config4_shard"], sec["fleet_config5"]
    for leg in (ik, rne, shard, fleet):
        assert "error" not in leg, leg
        assert leg["kernel_avg_ms"] > 0 and leg["value"] == pytest.approx(leg["n"] / (leg["kernel_avg_ms"] * 1e-3), rel=1e-9)
        assert 0.0 < leg["roofline"]["frac"] < 1.0
    assert ik["n"] == 100000 and 0.985 < ik["success_rate"] < 0.997
    assert ik["parity"]["same_success_iterations_searches"] == ik["parity"]["first_search_rows"] >= 200 and ik["parity"]["max_abs_dq"] <= 1e-6
    assert rne["n"] == 10000000 and rne["parity"]["max_rel_err"] <= 1e-9 and "frne" in rne["parity"]["against"]
    assert rne["roofline"]["achieved"] == pytest.approx(224e7 / (rne["kernel_avg_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert shard["n"] == 1250000
    assert fleet["n"] == 16000000 and fleet["parity"]["max_abs_err"] <= 1e-10 and len(fleet["arms"]) == 16
    # the FULL config 5: 17 chains of 16 robots, YuMi as the 14-DOF dual-arm robot on its 18-column q; 4..14 DOF; the 16-chain form beside it
    assert fleet["chains"] == 17 and fleet["arms"]["YuMi"] == 14 and min(fleet["arms"].values()) == 4 and max(fleet["arms"].values()) == 14
    assert fleet["parity"]["max_abs_err_yumi_arms"] <= 1e-10 and fleet["packed_layout_ms"] > 0 and fleet["packed_rows_equal_two_arrays"] is True
    assert fleet["sixteen_chain_form"]["kernel_avg_ms"] > 0 and 0.0 < fleet["sixteen_chain_form"]["roofline"]["frac"] < 1.0

"""-m "not gpu": the N>1 path (row sharding + the single output gather) with world_size 2 on the
gloo backend.  The compute kernel cannot run here, so each rank fills its shard with a function of
the GLOBAL row index; what is under test is the partition arithmetic (rtbhip_shard_range through
ShardedBatch) and the one collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, to_all, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
    import rtbhip
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sb = rtbhip.ShardedBatch(N)
        assert (sb.rank, sb.world) == (rank, world)
        rows = torch.arange(sb.begin, sb.begin + sb.count, dtype=torch.float64)
        local = torch.stack([rows * 3 + c for c in range(5)], dim=1)      # (count, 5), f(global row)
        full = sb.gather(local, to_all=to_all)
        if to_all or rank == 0:
            expect = torch.stack([torch.arange(N, dtype=torch.float64) * 3 + c for c in range(5)], dim=1)
            ok = full.shape == (N, 5) and bool(torch.equal(full, expect))
        else:
            ok = full is None
        q.put((rank, ok, sb.begin, sb.count))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("N,to_all", [(1000, True), (1001, True), (1001, False), (3, False), (64, True)])
def test_sharded_gather_world2(N, to_all):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N, to_all, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(ok for _, ok, _, _ in res)
    assert res[0][2] == 0 and res[0][2] + res[0][3] == res[1][2] and res[1][2] + res[1][3] == N


def test_single_process_is_identity():
    sys.path[:0] = [os.path.join(ROOT, "robotics-toolbox-python_amd")]
    import rtbhip
    sb = rtbhip.ShardedBatch(10, rank=0, world=1)
    x = torch.arange(10.0)
    assert sb.gather(x) is x and sb.local(x).shape == (10,)


def _bench_gather_worker(rank, world, port, N, q):
    """benchlib.Ranks.gather_ms -- the one exchange the benches time -- on two CPU ranks: the Ranks object is assembled by hand (its constructor
    insists on a GPU, as the benches must), backend gloo, host tensors."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
        import benchlib
        import rtbhip
        rk = object.__new__(benchlib.Ranks)
        rk.world, rk.rank, rk.local, rk.backend, rk.dist, rk.shared, rk.forced = world, rank, rank, "gloo", dist, False, False
        rk.dev = torch.device("cpu")
        begin, count = rtbhip.shard_range(N, rank, world)
        longest = max(rtbhip.shard_range(N, r, world)[1] for r in range(world))
        full = torch.arange(N * 7, dtype=torch.float64).reshape(N, 7)
        # the caller's buffer is as long as the longest shard (what a kernel wrote into); `rows` says how much of it is this rank's
        local = torch.full((longest, 7), -1.0, dtype=torch.float64)
        local[:count] = full[begin:begin + count]
        ms = rk.gather_ms(local, rows=count, keep=True)
        ok = ms is not None and ms >= 0.0 and torch.equal(rk.last_gather["rows"], full)
        ok = ok and rk.last_gather["buffer_bytes"] == world * longest * 56 and rk.last_gather["rows_per_rank_padded"] == longest
        q.put((rank, bool(ok), None))
        dist.destroy_process_group()
    except Exception as e:                       # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("N", [1001, 1000, 3, 1])
def test_bench_gather_ms_world2_uneven_shards(N):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_gather_worker, args=(r, 2, port, N, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, "rank %d: %s" % (rank, err)


def test_bench_gather_ms_world8_uneven_shards():
    """the shape of the run the driver does on an 8-GPU node: eight ranks, N not a multiple of eight"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_gather_worker, args=(r, 8, port, 1003, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _, _ in res) == list(range(8))
    for rank, ok, err in res:
        assert ok, "rank %d: %s" % (rank, err)


def _timed_steps_worker(rank, world, port, q):
    """benchlib.Ranks.timed_steps on host ranks whose steps take rank-dependent time: rank r sleeps (r + 1) ms per step."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
        import time
        import benchlib
        rk = object.__new__(benchlib.Ranks)
        rk.world, rk.rank, rk.local, rk.backend, rk.dist, rk.shared, rk.forced = world, rank, rank, "gloo", dist, False, False
        rk.dev = torch.device("cpu")
        K = 10
        elapsed, _ = rk.timed_steps(lambda: time.sleep(1e-3 * (rank + 1)), K, 1)
        q.put((rank, True, (elapsed, rk.last_own_elapsed, rk.last_elapsed_with_barrier)))
        dist.destroy_process_group()
    except Exception:                       # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc()))


def test_timed_region_has_no_collective_world8():
    """The figure `value` is made from is MAX over ranks of (t0 -> the rank's OWN synchronize); the closing barrier comes after t1.  Eight host
    ranks whose steps take (rank + 1) ms: every rank's own clock is its own work (not the slowest rank's), the returned figure is the slowest
    rank's own clock on every rank, and the barrier-inclusive figure is kept beside it and is not smaller."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timed_steps_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, err in res:
        assert ok, "rank %d: %s" % (rank, err)
    res.sort()
    slowest_own = max(v[1] for _, _, v in res)
    for rank, _, (elapsed, own, with_barrier) in res:
        assert elapsed == pytest.approx(slowest_own, rel=1e-12)           # MAX over ranks of the OWN clocks, the same on every rank
        assert with_barrier >= elapsed                                    # the closing barrier is outside t1 - t0
        assert 10e-3 * (rank + 1) <= own < 10e-3 * (rank + 1) + 8e-3      # a fast rank's clock does not wait for the slow ranks
    assert res[0][2][1] < 0.5 * res[7][2][1]
    import benchlib
    import inspect
    body = inspect.getsource(benchlib.Ranks.timed_steps)
    t1 = body.index("elapsed = time.perf_counter() - t0")
    assert body.index("self.barrier()") < body.index("t0 = time.perf_counter()") < t1 < body.rindex("self.barrier()")
    assert "ms_per_step_with_barrier" in open(os.path.join(ROOT, "bench.py")).read()


def test_gather_buffer_lives_outside_the_timed_region():
    """Structure of the bench scripts, checked on their source: the receive buffer of the output gather is allocated inside Ranks.gather_ms,
    which bench.py calls after Ranks.timed_steps has returned `elapsed` and before the cpu_baseline / secondary legs (which start only after
    the call has released it).  (The -m gpu twin, tests/test_dist_gpu.py, checks torch.cuda.memory_allocated() around a real RCCL gather.)"""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    pos = {}
    for node in ast.walk(main):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute):
            pos.setdefault(node.func.attr, node.lineno)
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name):
            pos.setdefault(node.func.id, node.lineno)
    assert pos["timed_steps"] < pos["gather_ms"] < pos["cpu_baseline"] < pos["secondary"]
    import benchlib
    import inspect
    body = inspect.getsource(benchlib.Ranks.timed_steps)
    assert "all_gather" not in body and "torch.empty" not in body           # nothing of the gather is inside the timed loop
    g = inspect.getsource(benchlib.Ranks.gather_ms)
    assert "torch.empty" in g and "return ms" in g

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

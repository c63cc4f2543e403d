"""rtbhip_shard_gather and its communicator entry points (include/rtbhip.h; SURVEY 8b's export list, 8e): argument checking without a GPU (the
refusals are made before RCCL or the device is touched), the duplicate-device check of the multi-rank bench line on two gloo ranks, and -- on the
GPU -- a world-size-1 RCCL communicator made through the C ABI: gather to root, to all, the grouped send / receive form, against torch."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import rtbhip
from rtbhip import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_gather_refuses_bad_geometry_before_touching_a_device():
    lib = _lib.lib()
    buf = (C.c_double * 8)()
    p = C.cast(buf, C.c_void_p)
    err = lambda: lib.rtbhip_last_error().decode()
    # rank 1 of 3 holds 3 of 10 rows (ranks 0 gets 4), not 4
    assert lib.rtbhip_shard_gather(None, p, 4, 56, 10, 3, 1, 0, None, None) == -1 and "holds 3 of 10 rows" in err()
    assert lib.rtbhip_shard_gather(None, p, 3, 56, 10, 3, 1, 0, None, None) == -1 and "needs a communicator" in err()      # world > 1 without one
    assert lib.rtbhip_shard_gather(None, p, 10, 0, 10, 1, 0, 0, p, None) == -1            # row_bytes < 1
    assert lib.rtbhip_shard_gather(None, p, 10, 56, 10, 1, 1, 0, p, None) == -1           # rank outside the world
    assert lib.rtbhip_shard_gather(None, p, 10, 56, 10, 1, 0, 1, p, None) == -1           # root outside the world
    assert lib.rtbhip_shard_gather(None, None, 10, 56, 10, 1, 0, 0, p, None) == -1 and "NULL local" in err()
    assert lib.rtbhip_shard_gather(None, p, 10, 56, 10, 1, 0, 0, None, None) == -1 and "NULL output" in err()
    assert lib.rtbhip_shard_gather(None, None, 0, 56, 0, 1, 0, 0, None, None) == 0        # nothing to move
    h = C.c_void_p()
    assert lib.rtbhip_shard_comm_create(None, 2, 0, C.byref(h)) == -1
    assert lib.rtbhip_shard_comm_create(C.c_char_p(b"x" * 128), 2, 2, C.byref(h)) == -1
    assert lib.rtbhip_shard_comm_destroy(None) == 0
    assert lib.rtbhip_device_copy(p, p, 8, 7, None) == -1 and "kind" in err()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _identities_worker(rank, world, port, same_device, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sys.path[:0] = [ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd")]
        import benchlib
        import rtbhip as r2
        # no GPU here: the identity of "this rank's device" is supplied by the test (the real one is rtbhip_device_identity)
        r2._lib.device_identity = lambda device=None: {"device": 0 if same_device else rank, "pci_bus_id": "0000:%02x:00.0" % (5 if same_device else 5 + rank),
                                                       "uuid": "%032x" % (7 if same_device else 7 + rank)}
        rk = object.__new__(benchlib.Ranks)
        rk.world, rk.rank, rk.local, rk.dist, rk.shared, rk.forced = world, rank, rank, dist, False, False
        rk.backend = "nccl"                       # the label only: all_gather_object runs on the gloo group made above
        rk.dev = torch.device("cpu")
        try:
            box = rk.identities(ms_per_step_own=0.1 * (rank + 1), kernel_avg_ms=0.05)
            ok = (not same_device) and [b["rank"] for b in box] == list(range(world)) and len({b["uuid"] for b in box}) == world \
                and all({"device", "pci_bus_id", "uuid", "ms_per_step_own", "kernel_avg_ms"} <= set(b) for b in box)
            q.put((rank, ok, None))
        except SystemExit as e:
            q.put((rank, same_device and "two ranks report the same GPU" in str(e), None))
        dist.destroy_process_group()
    except Exception:                            # noqa: BLE001
        import traceback
        q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("world,same_device", [(2, False), (2, True), (8, False)])
def test_rank_identities_and_the_duplicate_device_refusal(world, same_device):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_identities_worker, args=(r, world, port, same_device, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r for r, _, _ in res) == list(range(world))
    for rank, ok, err in res:
        assert ok, "rank %d: %s" % (rank, err)


@pytest.mark.gpu
def test_gpu_world1_rccl_communicator_through_the_c_abi():
    """One rank, a REAL communicator (ncclCommInitRank through rtbhip_shard_comm_create): the packed T||J rows gathered to root, to all, and in
    the grouped send / receive form, each equal to the shard itself; librccl mapped; the communicator reports world 1 rank 0."""
    comm = rtbhip.Communicator(rtbhip.Communicator.new_id(), 1, 0)
    w, r, ver = comm.info()
    assert (w, r) == (1, 0) and ver > 20000
    assert any("librccl" in l for l in open("/proc/self/maps"))
    ets = rtbhip.models.Panda().ets()
    N = 4097
    q = torch.from_numpy(np.random.default_rng(3).uniform(-3, 3, (N, 7))).cuda()
    _, _, TJ = ets.fkine_jacob0(q, packed=True)
    sb = rtbhip.ShardedBatch(N, 0, 1)
    a = sb.gather_rccl(comm, TJ, root=0)
    b = sb.gather_rccl(comm, TJ, root=-1)
    rtbhip.tune("shard_p2p", 1)
    try:
        c = sb.gather_rccl(comm, TJ, root=0)
        d = sb.gather_rccl(comm, TJ, root=-1)
    finally:
        rtbhip.tune("shard_p2p", 0)
    torch.cuda.synchronize()
    for x in (a, b, c, d):
        assert x.data_ptr() != TJ.data_ptr() and bool((x == TJ).all())
    tau = torch.arange(7.0 * 1001, dtype=torch.float64, device="cuda").reshape(1001, 7)          # a 56-byte row (config 4's message)
    out = torch.empty_like(tau)
    assert comm.gather(tau, 1001, root=0, out=out) is out
    torch.cuda.synchronize()
    assert bool((out == tau).all())
    with pytest.raises(rtbhip.RtbHipError):
        comm.gather(tau[:1000], 1001, root=0)                                # not this rank's rtbhip_shard_range count
    # comm = NULL, world 1: a device copy, no RCCL
    lib = _lib.lib()
    out.zero_()
    _lib.check(lib.rtbhip_shard_gather(None, C.c_void_p(tau.data_ptr()), 1001, 56, 1001, 1, 0, 0, C.c_void_p(out.data_ptr()), _lib.current_stream_ptr()))
    torch.cuda.synchronize()
    assert bool((out == tau).all())
    ident = _lib.device_identity()
    assert ident["device"] == torch.cuda.current_device() and len(ident["uuid"]) == 32 and ":" in ident["pci_bus_id"]
    comm.destroy()


@pytest.mark.gpu
def test_gpu_device_memory_helpers_of_the_abi():
    lib = _lib.lib()
    p, s = C.c_void_p(), C.c_void_p()
    _lib.check(lib.rtbhip_device_alloc(0, 8 * 100, C.byref(p)))
    _lib.check(lib.rtbhip_stream_create(0, C.byref(s)))
    a = np.arange(100.0)
    b = np.zeros(100)
    _lib.check(lib.rtbhip_device_copy(p, a.ctypes.data_as(C.c_void_p), 800, 1, s))
    _lib.check(lib.rtbhip_device_copy(b.ctypes.data_as(C.c_void_p), p, 800, 2, s))
    _lib.check(lib.rtbhip_stream_sync(s))
    assert np.array_equal(a, b)
    _lib.check(lib.rtbhip_stream_destroy(s))
    _lib.check(lib.rtbhip_device_free(p))
    assert lib.rtbhip_device_alloc(99, 8, C.byref(p)) == -1

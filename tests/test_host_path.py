"""-m gpu: the RTBHIP_MEM_HOST boundary (csrc/hostpipe.cpp) -- host arrays streamed through the device in row chunks over
two persistent slots.  Whatever the chunking and whether the arrays are pageable or pinned, the results must be BIT-equal to
the device-pointer path (same kernels, same rows), for fkine / jacob / fkine_jacob / hessian and rne."""
import ctypes as C

import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import _lib
from rtbhip._lib import lib, check, host_ptr, MEM_HOST

pytestmark = pytest.mark.gpu


@pytest.fixture
def small_chunks():
    rtbhip.tune("host_chunk_kb", 2048)            # ~4000 Panda rows per chunk: every size below spans many chunks
    yield
    rtbhip.tune("host_chunk_kb", 32 * 1024)


@pytest.mark.parametrize("N", [1, 63, 4033, 4096, 20000, 100003])
def test_host_arrays_equal_device_tensors_whatever_the_chunking(N, small_chunks):
    import torch
    rng = np.random.default_rng(N)
    ets = rtbhip.models.Panda().ets()
    q = rng.uniform(-np.pi, np.pi, (N, 7))
    qt = torch.from_numpy(q).cuda()
    Td, Jd = ets.fkine_jacob0(qt)
    Th, Jh = ets.fkine_jacob0(q)
    nt.assert_array_equal(np.asarray(Th).reshape(N, 4, 4), Td.cpu().numpy().reshape(N, 4, 4))
    nt.assert_array_equal(np.asarray(Jh).reshape(N, 6, 7), Jd.cpu().numpy().reshape(N, 6, 7))
    nt.assert_array_equal(np.asarray(ets.eval(q)).reshape(N, 4, 4), Td.cpu().numpy().reshape(N, 4, 4))
    nt.assert_array_equal(np.asarray(ets.jacobe(q)).reshape(N, 6, 7), ets.jacobe(qt).cpu().numpy().reshape(N, 6, 7))
    if N <= 20000:
        nt.assert_array_equal(np.asarray(ets.hessian0(q)).reshape(N, 7, 6, 7), ets.hessian0(qt).cpu().numpy().reshape(N, 7, 6, 7))
    arm = rtbhip.models.DH.Panda()
    qd, qdd = rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
    tau_d = arm.rne(qt, torch.from_numpy(qd).cuda(), torch.from_numpy(qdd).cuda()).cpu().numpy().reshape(N, 7)
    nt.assert_array_equal(np.asarray(arm.rne(q, qd, qdd)).reshape(N, 7), tau_d)
    g = arm.gravload(q)                                       # qd = qdd = NULL inputs are skipped, not staged
    nt.assert_array_equal(np.asarray(g).reshape(N, 7), arm.gravload(qt).cpu().numpy().reshape(N, 7))


def test_pageable_and_pinned_outputs_through_the_raw_abi(small_chunks):
    """The same call with (a) plain NumPy result arrays -- staged through the slots' pinned buffers and copied out by the copy
    threads -- and (b) result arrays from rtbhip_host_alloc -- DMA endpoints themselves."""
    N = 50001
    rng = np.random.default_rng(3)
    ets = rtbhip.models.Panda().ets()
    q = rng.uniform(-np.pi, np.pi, (N, 7))
    Ta, Ja = np.empty((N, 4, 4)), np.empty((N, 6, 7))
    check(lib().rtbhip_fkine_jacob(ets._handle(), host_ptr(q), N, None, None, 0, host_ptr(Ta), host_ptr(Ja), MEM_HOST, None))
    Tb, Jb = _lib.host_empty((N, 4, 4)), _lib.host_empty((N, 6, 7))
    assert Tb.base is not None and not Tb.flags.owndata            # a view of a pinned block
    check(lib().rtbhip_fkine_jacob(ets._handle(), host_ptr(q), N, None, None, 0, host_ptr(Tb), host_ptr(Jb), MEM_HOST, None))
    nt.assert_array_equal(Ta, Tb)
    nt.assert_array_equal(Ja, Jb)
    qp = _lib.host_empty((N, 7))                                   # pinned input as well: no staging copy at all
    qp[...] = q
    Tc = _lib.host_empty((N, 4, 4))
    check(lib().rtbhip_fkine(ets._handle(), host_ptr(qp), N, None, None, host_ptr(Tc), MEM_HOST, None))
    nt.assert_array_equal(Tc, Ta)
    from oracle import oracle, chains
    assert np.abs(Ta[:500] - oracle.fkine(chains.panda_ets(), q[:500])).max() <= 1e-10


def test_pinned_blocks_are_cached_and_outlive_their_makers():
    ets = rtbhip.models.Panda().ets()
    q = np.random.default_rng(0).uniform(-1, 1, (40000, 7))
    T = ets.eval(q)                                   # 5 MB: a pinned block
    keep = T[123].copy()
    del ets
    import gc
    gc.collect()
    nt.assert_array_equal(T[123], keep)               # the array owns its block
    p = C.c_void_p()
    check(lib().rtbhip_host_alloc(3 << 20, C.byref(p)))
    first = p.value
    check(lib().rtbhip_host_free(p))
    check(lib().rtbhip_host_alloc(3 << 20, C.byref(p)))
    assert p.value == first                           # came back from the cache: nothing was pinned again
    check(lib().rtbhip_host_free(p))
    assert lib().rtbhip_host_free(C.c_void_p(12345)) != 0        # not one of ours: refused, nothing freed
    small = _lib.host_empty((10, 4, 4))
    assert small.flags.owndata                        # below 1 MB an ordinary array

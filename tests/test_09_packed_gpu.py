"""-m gpu: rtbhip_fkine_jacob_packed / rtbhip_fleet_fkine_jacob_packed on the device -- the (N, 16 + 6n) rows must be bit for bit the two-array
form's (same chain walk; only the LDS staging and the store stream differ), for register-tile and run-time-n chains, ragged sizes, base / tool,
both frames, host and device buffers, and at BASELINE configs[1]'s full size; the Panda rows against the oracle."""
import numpy as np
import numpy.testing as nt
import pytest
import torch

import rtbhip
from oracle import oracle, chains
from helpers import full_size
from test_packed_output import random_chain

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 4, 7, 8, 9, 10, 11, 16])
def test_gpu_packed_rows_equal_two_arrays(n):
    rng = np.random.default_rng(100 + n)
    e = random_chain(rng, n)
    base = chains.elementary("Rz", 0.3) @ chains.elementary("tx", 0.2)
    tool = chains.elementary("Ry", -0.4) @ chains.elementary("tz", 0.1)
    for N in (2, 15, 16, 17, 63, 64, 65, 1000, 4097):
        qh = rng.uniform(-3, 3, (N, n))
        for frame in (0, 1):
            T, J = e.fkine_jacob0(qh, base=base, tool=tool, frame=frame)
            _, _, TJ = e.fkine_jacob0(qh, base=base, tool=tool, frame=frame, packed=True)                 # host buffers (row pipeline)
            nt.assert_array_equal(TJ[:, :16].reshape(N, 4, 4), np.asarray(T))
            nt.assert_array_equal(TJ[:, 16:].reshape(N, 6, n), np.asarray(J))
            Td, Jd, TJd = e.fkine_jacob0(torch.from_numpy(qh).cuda(), base=base, tool=tool, frame=frame, packed=True)   # device buffers
            assert TJd.is_cuda and tuple(TJd.shape) == (N, 16 + 6 * n) and tuple(Td.shape) == (N, 4, 4) and tuple(Jd.shape) == (N, 6, n)
            nt.assert_array_equal(TJd.cpu().numpy(), TJ)
            nt.assert_array_equal(Td.cpu().numpy(), np.asarray(T))
            nt.assert_array_equal(Jd.cpu().numpy(), np.asarray(J))


def test_gpu_packed_full_size_config2():
    N = full_size(1000000)
    ets = rtbhip.models.Panda().ets()
    ch = chains.panda_ets()
    qh = np.random.default_rng(0).uniform(-np.pi, np.pi, (N, 7))
    q = torch.from_numpy(qh).cuda()
    T, J = ets.fkine_jacob0(q)
    Tp, Jp, TJ = ets.fkine_jacob0(q, packed=True)
    assert bool((Tp == T).all()) and bool((Jp == J).all())
    buf = torch.full_like(TJ, float("nan"))
    ets.fkine_jacob0(q, packed=True, out=buf)
    assert bool((buf == TJ).all())                                                 # every element written, twice the same
    sel = np.arange(0, N, 997)
    rows = TJ[torch.from_numpy(sel).cuda()].cpu().numpy()
    nt.assert_allclose(rows[:, :16].reshape(-1, 4, 4), oracle.fkine(ch, qh[sel]), atol=1e-10)
    nt.assert_allclose(rows[:, 16:].reshape(-1, 6, 7), oracle.jacob0(ch, qh[sel]), atol=1e-10)


def test_gpu_fleet_packed_equals_fleet():
    rng = np.random.default_rng(6)
    es = [random_chain(rng, n) for n in (3, 7, 9, 12, 1, 8, 10, 14)]
    Ns = (700, 64, 5, 1290, 1, 4097, 333, 65)
    qs = [torch.from_numpy(rng.uniform(-2, 2, (N, e.n))).cuda() for e, N in zip(es, Ns)]
    Ts, Js = rtbhip.fleet_fkine_jacob(es, qs)
    TJs = rtbhip.fleet_fkine_jacob_packed(es, qs)
    torch.cuda.synchronize()
    for e, T, J, TJ in zip(es, Ts, Js, TJs):
        N = T.shape[0]
        assert bool((TJ[:, :16].reshape(N, 4, 4) == T).all()) and bool((TJ[:, 16:].reshape(N, 6, e.n) == J).all())
    hq = [q.cpu().numpy() for q in qs]
    hTJ = rtbhip.fleet_fkine_jacob_packed(es, hq)                                   # host buffers
    for a, b in zip(hTJ, TJs):
        nt.assert_array_equal(a, b.cpu().numpy())

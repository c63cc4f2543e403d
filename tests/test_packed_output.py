"""rtbhip_fkine_jacob_packed / rtbhip_fleet_fkine_jacob_packed: one (N, 16 + 6n) array of [T | J] rows (SURVEY 8e's T||J gather message; a
single write stream).  The rows must be BIT FOR BIT what rtbhip_fkine_jacob writes into its two arrays -- same chain walk, only the staging
differs -- for every joint count (register tile 1..10, run-time-n tile beyond), ragged batch sizes, a base, a tool, both frames.
Here (no GPU) through the kernel bodies' CPU replay; tests/test_00_gpu_parity.py has the device twin."""
import numpy as np
import pytest

import cpu_backend
import rtbhip
from oracle import chains, oracle


def random_chain(rng, n):
    ets = []
    for j in range(n):
        ets.append(rtbhip.ET.SE3(chains.random_se3(rng)) if hasattr(chains, "random_se3") else rtbhip.ET.tx(float(rng.normal())))
        ets.append(getattr(rtbhip.ET, str(rng.choice(["Rx", "Ry", "Rz", "tx", "ty", "tz"])))(flip=bool(rng.integers(2))))
    ets.append(rtbhip.ET.tz(0.1))
    return rtbhip.ETS(ets)


@pytest.mark.parametrize("n", [1, 2, 5, 7, 8, 10, 11, 14])
def test_packed_rows_equal_the_two_array_form(n):
    rng = np.random.default_rng(n)
    with cpu_backend.installed() as be:
        e = random_chain(rng, n)
        base = chains.elementary("Rz", 0.3) @ chains.elementary("tx", 0.2)
        tool = chains.elementary("Ry", -0.4) @ chains.elementary("tz", 0.1)
        for N in (1, 15, 16, 17, 63, 64, 65, 130):
            q = rng.uniform(-3, 3, (N, n))
            for frame in (0, 1):
                T, J = e.fkine_jacob0(q, base=base, tool=tool, frame=frame)
                Tp, Jp, TJ = e.fkine_jacob0(q, base=base, tool=tool, frame=frame, packed=True)
                np.testing.assert_array_equal(np.asarray(Tp), np.asarray(T))
                np.testing.assert_array_equal(np.asarray(Jp), np.asarray(J))
                TJ = np.asarray(TJ).reshape(N, -1)                      # (a (1, n) q is ONE configuration, as in the reference: 1-D row)
                assert TJ.shape == (N, 16 + 6 * n)
                np.testing.assert_array_equal(TJ[:, :16].reshape(N, 4, 4), np.asarray(T).reshape(N, 4, 4))
                np.testing.assert_array_equal(TJ[:, 16:].reshape(N, 6, n), np.asarray(J).reshape(N, 6, n))
        assert be.calls.get("rtbhip_fkine_jacob_packed", 0) >= 16
        # a single configuration: (4,4), (6,n) views and the 1-D row
        T1, J1, r1 = e.fkine_jacob0(q[0], packed=True)
        assert T1.shape == (4, 4) and J1.shape == (6, n) and r1.shape == (16 + 6 * n,)


def test_packed_panda_against_the_oracle_and_out_buffer():
    with cpu_backend.installed():
        e = rtbhip.models.Panda().ets()
        ch = chains.panda_ets()
        q = np.random.default_rng(0).uniform(-np.pi, np.pi, (200, 7))
        buf = np.full((200, 58), np.nan)
        T, J, TJ = e.fkine_jacob0(q, packed=True, out=buf)
        assert TJ is buf and not np.isnan(buf).any()
        assert np.abs(np.asarray(T) - oracle.fkine(ch, q)).max() < 1e-12
        assert np.abs(np.asarray(J) - oracle.jacob0(ch, q)).max() < 1e-12
        with pytest.raises(ValueError):
            e.fkine_jacob0(q, packed=True, out=np.zeros((200, 57)))


def test_packed_refusals_are_the_two_array_form_s():
    with cpu_backend.installed():
        e = rtbhip.models.Panda().ets()
        L = rtbhip._lib
        TJ = np.zeros((4, 58))
        q = np.zeros((4, 7))
        rc = L.lib().rtbhip_fkine_jacob_packed(e._handle(), L.host_ptr(q), 4, None, None, 2, L.host_ptr(TJ), L.MEM_HOST, None)
        assert rc == -1 and b"frame" in L.lib().rtbhip_last_error()
        rc = L.lib().rtbhip_fkine_jacob_packed(e._handle(), L.host_ptr(q), 4, None, None, 0, None, L.MEM_HOST, None)
        assert rc == -1 and b"no output buffer" in L.lib().rtbhip_last_error()
        assert L.lib().rtbhip_fkine_jacob_packed(e._handle(), None, 0, None, None, 0, None, L.MEM_HOST, None) == 0


def test_fleet_packed_equals_fleet():
    rng = np.random.default_rng(5)
    with cpu_backend.installed():
        es = [random_chain(rng, n) for n in (3, 7, 9, 12, 1)]
        qs = [rng.uniform(-2, 2, (N, e.n)) for e, N in zip(es, (70, 64, 5, 129, 1))]
        Ts, Js = rtbhip.fleet_fkine_jacob(es, qs)
        TJs = rtbhip.fleet_fkine_jacob_packed(es, qs)
        for e, T, J, TJ in zip(es, Ts, Js, TJs):
            N = T.shape[0]
            np.testing.assert_array_equal(TJ[:, :16].reshape(N, 4, 4), T)
            np.testing.assert_array_equal(TJ[:, 16:].reshape(N, 6, e.n), J)
        again = rtbhip.fleet_fkine_jacob_packed(es, qs, out=TJs)
        assert all(a is b for a, b in zip(again, TJs))

"""-m "not gpu": runs the kernels' own per-lane phase functions on the CPU (tests/emu) and checks
them against the oracle and the golden fixtures: same LDS layout, same flush arithmetic, same
recursions as the gfx950 kernels -- only the lanes are a for-loop."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
import emu_harness as emu
from oracle import oracle, chains
from helpers import literals, ref_outputs, mixed_spec, product_ets, tool_base

LIT = literals()
REF = ref_outputs()


def test_sincos_accuracy_and_fallback():
    """trig.h: branch-free reduced-range path within 5e-16 of libm for |x| < 2^20, library beyond."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-np.pi, np.pi, 200000), rng.uniform(-1e3, 1e3, 200000),
                        rng.uniform(-1048575, 1048575, 200000), np.linspace(-7, 7, 100001),
                        np.array([0.0, -0.0, np.pi / 2, np.pi, -np.pi, 1e-300, 1048575.9])])
    s, c = emu.sincos(x, reduced_only=True)
    assert np.abs(s - np.sin(x)).max() < 5e-16 and np.abs(c - np.cos(x)).max() < 5e-16
    xb = np.array([1048576.0, 1e7, -3e9, 2.0 ** 40 + 0.5, 1e300])
    s, c = emu.sincos(xb)
    nt.assert_array_equal(s, np.sin(xb))
    nt.assert_array_equal(c, np.cos(xb))
    s, c = emu.sincos(np.array([np.nan, np.inf, -np.inf]))
    assert np.isnan(s).all() and np.isnan(c).all()


@pytest.mark.parametrize("reg", [True, False])
def test_register_variant_fixture_parity(reg):
    """kin_reg.h (compile-time joint count, n <= 8) against the reference-run fixtures."""
    tool, base = tool_base()
    ets = rtbhip.models.Panda().ets()
    q = REF["panda_q"]
    T, J, _ = emu.kin(ets, q, reg=reg)
    nt.assert_allclose(T, REF["panda_fkine"], atol=1e-12)
    nt.assert_allclose(J, REF["panda_jacob0"], atol=1e-12)
    T, J, _ = emu.kin(ets, q, base=base, tool=tool, reg=reg)
    nt.assert_allclose(T, REF["panda_fkine_bt"], atol=1e-12)
    nt.assert_allclose(J, REF["panda_jacob0_tool"], atol=1e-12)
    _, J, _ = emu.kin(ets, q, tool=tool, frame=1, want=("J",), reg=reg)
    nt.assert_allclose(J, REF["panda_jacobe_tool"], atol=1e-12)
    T, _, _ = emu.kin(ets, q, base=base, want=("T",), reg=reg)
    nt.assert_allclose(T, oracle.fkine(chains.panda_ets(), q, base=base), atol=1e-12)
    mx = product_ets(mixed_spec())
    for frame, key in ((0, "mixed_jacob0"), (1, "mixed_jacobe")):
        T, J, _ = emu.kin(mx, REF["mixed_q"], frame=frame, reg=reg)
        nt.assert_allclose(T, REF["mixed_fkine"], atol=1e-12)
        nt.assert_allclose(J, REF[key], atol=1e-12)


@pytest.mark.parametrize("N", [1, 31, 32, 33, 63, 64, 65, 97, 1000])
def test_register_variant_ragged_tiles(N):
    ets = rtbhip.models.Panda().ets()
    ch = chains.panda_ets()
    rng = np.random.default_rng(N)
    q = rng.uniform(-np.pi, np.pi, (N, 7))
    T, J, _ = emu.kin(ets, q, reg=True)
    nt.assert_allclose(T, oracle.fkine(ch, q), atol=1e-12)
    nt.assert_allclose(J, oracle.jacob0(ch, q), atol=1e-12)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 8])
def test_register_variant_joint_counts(n):
    rng = np.random.default_rng(40 + n)
    axes = ["Rx", "Ry", "Rz", "tx", "ty", "tz"]
    spec = []
    for j in range(n):
        if rng.integers(3):
            spec.append((axes[rng.integers(6)], float(rng.normal())))
        spec.append((axes[rng.integers(6)], None, bool(rng.integers(2))))
    ets = product_ets(spec)
    ch = chains.Chain(spec)
    q = rng.normal(size=(70, n))
    for frame in (0, 1):
        T, J, _ = emu.kin(ets, q, frame=frame, reg=True)
        nt.assert_allclose(T, oracle.fkine(ch, q), atol=1e-11)
        nt.assert_allclose(J, oracle.jacob(ch, q, frame=frame), atol=1e-11)


@pytest.mark.parametrize("coalesced", [True, False])
def test_panda_fixture_parity(coalesced):
    tool, base = tool_base()
    ets = rtbhip.models.Panda().ets()
    q = REF["panda_q"]
    T, J, H = emu.kin(ets, q, want=("T", "J", "H"), coalesced=coalesced)
    nt.assert_allclose(T, REF["panda_fkine"], atol=1e-12)
    nt.assert_allclose(J, REF["panda_jacob0"], atol=1e-12)
    nt.assert_allclose(H[:8], REF["panda_hessian0"], atol=1e-12)
    T, J, _ = emu.kin(ets, q, base=base, tool=tool, coalesced=coalesced)
    nt.assert_allclose(T, REF["panda_fkine_bt"], atol=1e-12)
    nt.assert_allclose(J, REF["panda_jacob0_tool"], atol=1e-12)
    _, J, _ = emu.kin(ets, q, tool=tool, frame=1, want=("J",), coalesced=coalesced)
    nt.assert_allclose(J, REF["panda_jacobe_tool"], atol=1e-12)


def test_golden_literals_through_kernel_body():
    ets = rtbhip.models.Panda().ets()
    T, J, H = emu.kin(ets, LIT["panda_q"], want=("T", "J", "H"))
    nt.assert_array_almost_equal(T[0], LIT["G1_panda_fkine"], decimal=6)
    nt.assert_array_almost_equal(J[0], LIT["G2_panda_jacob0"], decimal=6)
    raw = LIT["G8_panda_hessian0_raw"]
    nt.assert_array_almost_equal(H[0], np.stack([raw[:, :, i] for i in range(7)]), decimal=6)


def test_mixed_chain_every_axis_flip_and_se3():
    ets = product_ets(mixed_spec())
    q = REF["mixed_q"]
    for frame, key in ((0, "mixed_jacob0"), (1, "mixed_jacobe")):
        T, J, _ = emu.kin(ets, q, frame=frame)
        nt.assert_allclose(T, REF["mixed_fkine"], atol=1e-12)
        nt.assert_allclose(J, REF[key], atol=1e-12)


@pytest.mark.parametrize("N", [1, 63, 64, 65, 129, 1000])
def test_ragged_tiles(N):
    ets = rtbhip.models.Panda().ets()
    ch = chains.panda_ets()
    rng = np.random.default_rng(N)
    q = rng.uniform(-np.pi, np.pi, (N, 7))
    T, J, _ = emu.kin(ets, q)
    nt.assert_allclose(T, oracle.fkine(ch, q), atol=1e-12)
    nt.assert_allclose(J, oracle.jacob0(ch, q), atol=1e-12)


@pytest.mark.parametrize("n", [1, 2, 3, 5, 11, 16, 23])
def test_joint_counts_from_1_to_23(n):
    rng = np.random.default_rng(n)
    axes = ["Rx", "Ry", "Rz", "tx", "ty", "tz"]
    spec = []
    for j in range(n):
        spec.append((axes[rng.integers(6)], float(rng.normal())))
        spec.append((axes[rng.integers(6)], None, bool(rng.integers(2))))
    ets = product_ets(spec)
    ch = chains.Chain(spec)
    q = rng.normal(size=(70, n))
    for frame in (0, 1):
        T, J, _ = emu.kin(ets, q, frame=frame)
        nt.assert_allclose(T, oracle.fkine(ch, q), atol=1e-11)
        nt.assert_allclose(J, oracle.jacob(ch, q, frame=frame), atol=1e-11)


def test_explicit_jindex_permutation():
    e = rtbhip.ET.Rz(jindex=2) * rtbhip.ET.tx(0.5) * rtbhip.ET.Ry(jindex=0) * rtbhip.ET.tz(jindex=1)
    rng = np.random.default_rng(5)
    q = rng.normal(size=(10, 3))
    ch = chains.Chain([("Rz",), ("tx", 0.5), ("Ry",), ("tz",)])
    ch.jindex = np.array([2, 0, 0, 1], dtype=np.int32)
    T, J, _ = emu.kin(e, q)
    nt.assert_allclose(T, oracle.fkine(ch, q), atol=1e-12)
    nt.assert_allclose(J, oracle.jacob0(ch, q), atol=1e-12)   # columns in chain order (methods.cpp:120,201)


def test_rne_fixture_parity():
    pu, pd = chains.puma560(), chains.panda_dh()
    for generic in (False, True):
        tau = emu.rne(pu.L24(), 0, REF["puma_q"], REF["puma_qd"], REF["puma_qdd"], -pu.gravity, [1, 2, 3, 1, 2, 3], generic)
        nt.assert_allclose(tau, REF["puma_rne_fext"], rtol=1e-11, atol=1e-11)
        tau = emu.rne(pd.L24(), 1, REF["pandadh_q"], REF["pandadh_qd"], REF["pandadh_qdd"], -pd.gravity, None, generic)
        nt.assert_allclose(tau, REF["pandadh_rne"], rtol=1e-11, atol=1e-11)
        for name in ("rprp0", "rprp1", "prp0", "prp1"):
            fext = [1, 2, 3, 4, 5, 6] if name.startswith("rprp") else None
            tau = emu.rne(REF[name + "_L24"], int(name[-1]), REF[name + "_q"], REF[name + "_qd"], REF[name + "_qdd"],
                          -np.array([0.5, -1.0, -9.81]), fext, generic)
            nt.assert_allclose(tau, REF[name + "_rne"], rtol=1e-11, atol=1e-11)


def test_rne_golden_literals_through_kernel_body():
    pu = chains.puma560()
    z, o = np.zeros(6), np.ones(6)
    tau = emu.rne(pu.L24(), 0, chains.PUMA_QN, o, o, -pu.gravity)
    nt.assert_array_almost_equal(tau[0], LIT["G9_puma_rne_tr2"], decimal=4)
    tau = emu.rne(pu.L24(), 0, chains.PUMA_QN, z, z, -pu.gravity, LIT["G9_fext"])
    nt.assert_array_almost_equal(tau[0], LIT["G9_puma_rne_tr5"], decimal=4)

"""-m "not gpu": runs the kernels' own per-lane phase functions on the CPU (tests/emu) and checks
them against the oracle and the golden fixtures: same LDS layout, same flush arithmetic, same
recursions as the gfx950 kernels -- only the lanes are a for-loop."""
import os
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


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 8, 9, 10])
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


def test_branch_chain_on_robot_wide_q():
    """q_width beyond max(jindex)+1 (rtbhip_chain_set_q_width): YuMi's two arms read the same 18-column rows and give what the
    path-numbered chains give on the picked columns -- LDS-tile walk and register walk."""
    from rtbhip import urdf
    y = urdf.load("YuMi")
    rng = np.random.default_rng(18)
    q = rng.uniform(-1.0, 1.0, (130, y.n))
    for end in ("gripper_r_finger_r", "gripper_l_finger_l"):
        wide, local = y.ets(end=end, compact=False), y.ets(end=end)
        assert wide.q_width == 18 and local.q_width == local.n == 8
        qs = np.ascontiguousarray(q[:, wide.jindices])
        for reg in (False, True):
            Tw, Jw, _ = emu.kin(wide, q, reg=reg)
            Tl, Jl, _ = emu.kin(local, qs, reg=reg)
            nt.assert_array_equal(Tw, Tl)
            nt.assert_array_equal(Jw, Jl)


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


# ---------------------------------------------------------------- inverse kinematics
def _panda_limited():
    ets = rtbhip.models.Panda().ets()
    ets.qlim = chains.PANDA_QLIM
    return ets, chains.panda_ets(with_limits=True)


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("method,k", [("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)])
def test_ik_state_machine_equals_oracle_loops(flavour, method, k):
    """Fed the same restart vectors, the per-lane state machine (ik_device.h) must walk exactly the
    reference's nested loops: identical (success, iterations, searches), q to 1e-6."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(17)
    N = 40
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    q, ok, it, se, E = emu.ik(ets, Tep, k=k, method=method, flavour=flavour, seed=99)
    for i in range(N):
        rs = np.array([emu.ik_restart(ets, 99, i, d) for d in range(101)])
        o = oracle.ik_lm(ch, Tep[i], k=k, method=method, restarts=rs) if flavour == 0 else \
            oracle.ikine_lm(ch, Tep[i], rs[:100], k=k, method=method)
        assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
        nt.assert_allclose(q[i], o[0], atol=1e-6)
        assert abs(E[i] - o[4]) <= 1e-9 * max(1.0, abs(o[4]))
    assert ok.mean() > 0.9


def test_ik_reference_run_fixtures_first_search():
    """q0 supplied and converging in the first search: no RNG involved, must match the reference's
    own IK_LM_c output (tests/golden/ref_outputs.npz) -- SURVEY 8c."""
    ets, _ = _panda_limited()
    for method, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):
        q, ok, it, se, E = emu.ik(ets, REF["ik_Tep"], q0=REF["ik_q0"], k=k, method=method)
        meta = REF["ik_%s_meta" % method]
        first = (meta[:, 2] == 1) & (meta[:, 0] == 1)
        assert first.sum() >= 15
        nt.assert_array_equal(np.c_[ok, it, se][first], meta[first])
        nt.assert_allclose(q[first], REF["ik_%s_q" % method][first], atol=1e-6)


def test_ik_golden_G10_and_mask_and_limits():
    ets, ch = _panda_limited()
    qr = np.array([0, -0.3, 0, -2.2, 0, 2, np.pi / 4])
    Tep = oracle.fkine(ch, qr)
    for method, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):      # test_IK.py:632-708
        q, ok, it, se, E = emu.ik(ets, Tep, k=k, method=method, seed=3)
        assert ok[0] == 1
        e = oracle.angle_axis(oracle.fkine(ch, q[0])[0], Tep[0])
        assert 0.5 * e @ e < 1e-5
        assert np.all(q[0] >= ch.qlim[0]) and np.all(q[0] <= ch.qlim[1])
    # position-only mask: orientation is free, the position must be met
    q, ok, it, se, E = emu.ik(ets, Tep, mask=[1, 1, 1, 0, 0, 0], seed=4)
    assert ok[0] == 1 and np.abs(oracle.fkine(ch, q[0])[0][:3, 3] - Tep[0][:3, 3]).max() < 2e-3
    # an unreachable target exhausts the searches: (success, searches) as the reference reports them
    far = Tep.copy(); far[0, :3, 3] = [3.0, 3.0, 3.0]
    q, ok, it, se, E = emu.ik(ets, far, ilimit=5, slimit=4, seed=5)
    assert ok[0] == 0 and se[0] == 5 and it[0] == 6 + 3 * 6      # ik.cpp:39,66-68: 1..5 then 0..5 three times
    q, ok, it, se, E = emu.ik(ets, far, ilimit=5, slimit=4, seed=5, flavour=1)
    assert ok[0] == 0 and se[0] == 4 and it[0] == 20               # IK.py:311-366


def test_ik_restart_generator_is_uniform_in_limits():
    ets, ch = _panda_limited()
    r = np.array([emu.ik_restart(ets, 7, t, d) for t in range(200) for d in range(10)])
    assert np.all(r >= ch.qlim[0]) and np.all(r < ch.qlim[1])
    u = (r - ch.qlim[0]) / (ch.qlim[1] - ch.qlim[0])
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.std() - 12 ** -0.5) < 0.02
    assert len(np.unique(r)) == r.size


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("N,waves,slimit,with_q0", [(300, 3, 100, False), (64, 1, 100, False), (130, 4, 7, True),
                                                     (1, 2, 100, False), (500, 2, 40, False)])
def test_ik_wave_scheduler_equals_sequential_searches(flavour, N, waves, slimit, with_q0):
    """The per-wave speculative scheduler (fresh targets, then parallel later searches, accounted in
    search order) must report exactly what running each target's searches one after another reports,
    including for unreachable targets that exhaust every search and joint-limit rejections."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(23 + N)
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    Tep[::17, :3, 3] += 2.5                     # some unreachable targets: every search fails
    q0 = rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)) if with_q0 else None
    a = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=11)
    st = [0, 0, 0, 0]
    b = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=11, waves=waves, stats=st)
    for x, y in zip(a, b):
        nt.assert_array_equal(x, y)
    import os
    for mask in ("1", "3"):                      # the pass may run only every 2nd / 4th iteration: same outputs
        os.environ["EMU_IK_PASS_MASK"] = mask
        try:
            c = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=11, waves=waves)
        finally:
            del os.environ["EMU_IK_PASS_MASK"]
        for x, y in zip(a, c):
            nt.assert_array_equal(x, y)
    assert a[1].sum() < N                        # the failures really are in the mix
    assert st[2] >= a[2].sum() - N * 2           # speculation only ever ADDS lane-iterations


@pytest.mark.parametrize("N", [1, 63, 64, 65, 200])
def test_hessian_register_path_equals_oracle_and_lds_tile_path(N):
    """k_kin_hess: 64 staged Jacobians -> every lane generates 16-byte pieces of the tile's contiguous
    (N,n,6,n) run.  Must equal the oracle (methods.cpp:16-32) and the run-time-n LDS-tile path, for both
    frames, odd and even joint counts (odd n*6*n run lengths end on a single double)."""
    rng = np.random.default_rng(N)
    for ets, ch in ((rtbhip.models.Panda().ets(), chains.panda_ets()),
                    (rtbhip.models.DH.Puma560().ets(), chains.puma560().ets())):
        q = rng.uniform(-2, 2, (N, ets.n))
        for frame in (0, 1):
            Hr = emu.hess_reg(ets, q, frame=frame)
            nt.assert_allclose(Hr, oracle.hessian(ch, q, frame=frame), atol=1e-12)
            _, _, Ht = emu.kin(ets, q, frame=frame, want=("H",))
            nt.assert_allclose(Hr, Ht, atol=1e-13)
            for rounds in (4, 8, 16):                               # k_kin_hess_tile<NJ, R>
                nt.assert_array_equal(emu.hess_reg(ets, q, frame=frame, rounds=rounds), Hr)
    three = rtbhip.ET.Rz() * rtbhip.ET.tx(0.3) * rtbhip.ET.Ry() * rtbhip.ET.tz(0.2) * rtbhip.ET.tx()
    q3 = rng.uniform(-1, 1, (N, 3))
    _, _, Ht = emu.kin(three, q3, want=("H",))
    nt.assert_allclose(emu.hess_reg(three, q3), Ht, atol=1e-13)
    nt.assert_allclose(emu.hess_reg(three, q3, rounds=8), Ht, atol=1e-13)


@pytest.mark.parametrize("n", [1, 2, 5, 7, 10, 11, 14])
def test_hessian_from_supplied_jacobian(n):
    """k_hess_from_jac (ETS_hessian0 / ETS_hessiane with J given, fknm.cpp:583-783 -> methods.cpp:16-32): Jacobian tiles
    through LDS, register expansion, tile flush; the run-time-n lane loop beyond 10 joints.  Against the oracle's Hessian
    of the same chain, both frames, ragged batch sizes."""
    spec = []
    rng = np.random.default_rng(n)
    for j in range(n):
        spec.append((["Rz", "Ry", "Rx", "tz", "tx"][j % 5], None, bool(j % 3 == 2)))
        spec.append((["tx", "tz", "Ry"][j % 3], float(rng.uniform(-0.4, 0.4))))
    ch = chains.Chain(spec, name="c%d" % n)
    for N in (1, 63, 64, 65, 130):
        q = rng.uniform(-2, 2, (N, n))
        for frame in (0, 1):
            J = oracle.jacob(ch, q, None, frame)
            H = emu.hess_from_jac(J)
            nt.assert_allclose(H, oracle.hessian(ch, q, frame=frame), atol=1e-13)


def test_ik_gn_nr_reference_run_fixtures_first_search():
    """IK_GN_c / IK_NR_c (the reference's own extension, tests/golden/ref_outputs.npz): where the first
    search converges no RNG is involved and the minimum-norm step equals the reference's SVD / QR / damped
    pseudo-inverse step to rounding: same (success, iterations, searches), q to 1e-6."""
    ets, _ = _panda_limited()
    puma = rtbhip.models.DH.Puma560().ets()
    puma.qlim = chains.puma560().qlim.T
    n_checked = 0
    for key, e, Tep, q0, kw in (("ik_gn", ets, REF["ik_Tep"], REF["ik_q0"], dict(method="gn")),
                                ("ik_nr", ets, REF["ik_Tep"], REF["ik_q0"], dict(method="nr", k=0.0)),
                                ("ik_nrd", ets, REF["ik_Tep"], REF["ik_q0"], dict(method="nr", k=0.05)),
                                ("ik6_gn", puma, REF["ik6_Tep"], REF["ik6_q0"], dict(method="gn")),
                                ("ik6_nr", puma, REF["ik6_Tep"], REF["ik6_q0"], dict(method="nr", k=0.0))):
        q, ok, it, se, E = emu.ik(e, Tep, q0=q0, **kw)
        meta = REF[key + "_meta"]
        first = (meta[:, 2] == 1) & (meta[:, 0] == 1)
        n_checked += int(first.sum())
        nt.assert_array_equal(np.c_[ok, it, se][first], meta[first])
        nt.assert_allclose(q[first], REF[key + "_q"][first], atol=1e-6)
    assert n_checked >= 40


@pytest.mark.parametrize("step", ["nr", "gn"])
def test_ikine_nr_gn_python_loop_semantics(step):
    """ikine_NR / ikine_GN (flavour 1 + pseudo-inverse step) against a NumPy restatement of IKSolver._solve with
    numpy.linalg.pinv.  Undamped Newton steps from random restarts pass near singular configurations where an
    SVD-truncated pseudo-inverse and an exact minimum-norm solve part ways, so the comparison is made where no
    restart is involved: start vectors near the solution, first search converges -> same iteration count, q to 1e-6."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(31)
    N = 24
    qs = rng.uniform(ch.qlim[0] + 0.15, ch.qlim[1] - 0.15, (N, 7))
    Tep = oracle.fkine(ch, qs)
    q0 = qs + 0.05 * rng.normal(size=qs.shape)
    q, ok, it, se, E = emu.ik(ets, Tep, q0=q0, method=step, flavour=1, seed=77, slimit=5, k=0.0)   # k carries pinv_damping
    checked = 0
    for i in range(N):
        rs = np.array([q0[i]] + [emu.ik_restart(ets, 77, i, d) for d in range(1, 5)])
        o = oracle.ikine_py(ch, Tep[i], rs, step=step, slimit=5)
        if o[1] and o[3] == 1:
            checked += 1
            assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
            nt.assert_allclose(q[i], o[0], atol=1e-6)
    assert checked >= 12


@pytest.mark.parametrize("step,method,k,ns", [("lm", "chan", 1.0, (0.1, 0.1, 0.0, 0.3)), ("lm", "sugihara", 0.01, (0.5, 0.0, 0.05, 0.4)),
                                              ("nr", "nr", 0.0, (0.01, 1.0, 0.0, 0.3)), ("gn", "gn", 0.0, (1.0, 1.0, 0.0, 0.3))])
def test_ikine_nullspace_terms_equal_python_solver(step, method, k, ns):
    """kq / km / ps / pi of IK_LM / IK_NR / IK_GN (robot/IK.py:507-576, added at :758, :1015, :1215): the device
    step against the NumPy restatement (numpy.linalg.pinv projector, oracle jacobm), first-search cases started near
    the solution and near a joint limit so the avoidance term is active.  Parameter sets of tests/test_IK.py:166-173
    (NR kq=0.01 km=1), :194-196 (LM kq=km=0.1), :261-263 (GN kq=km=1)."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(41)
    N = 20
    qs = rng.uniform(ch.qlim[0] + 0.25, ch.qlim[1] - 0.25, (N, 7))
    qs[::2, 3] = ch.qlim[1, 3] - 0.12                      # inside the influence distance of joint 4's upper limit
    qs[1::4, 1] = ch.qlim[0, 1] + 0.1
    Tep = oracle.fkine(ch, qs)
    q0 = qs + 0.03 * rng.normal(size=qs.shape)
    q0 = np.clip(q0, ch.qlim[0] + 0.02, ch.qlim[1] - 0.02)
    kq, km, ps, pi = ns
    emu.ik_nullspace(kq, km, ps, pi)
    try:
        q, ok, it, se, E = emu.ik(ets, Tep, q0=q0, method=method, flavour=1, seed=5, slimit=3, k=k)
        qw, okw, itw, sew, Ew = emu.ik(ets, Tep, q0=q0, method=method, flavour=1, seed=5, slimit=3, k=k, waves=2)
    finally:
        emu.ik_nullspace()
    nt.assert_array_equal(q, qw)                            # the wave scheduler replays the same searches
    nt.assert_array_equal(np.c_[ok, it, se], np.c_[okw, itw, sew])
    # the term must have changed something: without it the same problem takes a different path
    q_plain = emu.ik(ets, Tep, q0=q0, method=method, flavour=1, seed=5, slimit=3, k=k)[0]
    assert np.nanmax(np.abs(q_plain - q)) > 1e-6
    checked = 0
    for i in range(N):
        rs = np.array([q0[i]] + [emu.ik_restart(ets, 5, i, d) for d in range(1, 3)])
        o = oracle.ikine_py(ch, Tep[i], rs, step=step, slimit=3, k=k, kq=kq, km=km, ps=ps, pi=pi, method=method if step == "lm" else "chan")
        if o[1] and o[3] == 1:
            checked += 1
            assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
            nt.assert_allclose(q[i], o[0], atol=1e-7)
    assert checked >= 10


@pytest.mark.parametrize("robot", ["AL5D", "px100"])
def test_ik_gn_nr_on_arms_with_fewer_than_six_joints(robot):
    """Gauss-Newton / Newton-Raphson on 4- and 5-joint arms: the 6 x n Jacobian has full COLUMN rank, J J^T is singular, and
    the pseudo-inverse step is the least-squares one (n x n normal equations).  Against the NumPy restatement with
    numpy.linalg.pinv (first-search cases: same counts, q to 1e-6) and, where the reference's own build is present, against
    IK_GN_c / IK_NR_c themselves; the success rate must be that of the pseudo-inverse, not of a noise-pivot solve."""
    from rtbhip import urdf
    from helpers import chain_from_ets
    from oracle import ref_harness
    ets = urdf.load(robot).ets()
    assert ets.n < 6
    ets.qlim = np.clip(ets.qlim, -np.pi, np.pi)
    ch = chain_from_ets(ets)
    rng = np.random.default_rng(ets.n)
    N = 40
    qs = rng.uniform(ch.qlim[0] + 0.1, ch.qlim[1] - 0.1, (N, ets.n))
    Tep = oracle.fkine(ch, qs)
    q0 = qs + 0.1 * rng.normal(size=qs.shape)
    for method in ("gn", "nr"):
        q, ok, it, se, E = emu.ik(ets, Tep, q0=q0, method=method, flavour=1, seed=3, slimit=10, k=0.0)
        checked = 0
        for i in range(N):
            rs = np.array([q0[i]] + [emu.ik_restart(ets, 3, i, d) for d in range(1, 10)])
            o = oracle.ikine_py(ch, Tep[i], rs, step=method, slimit=10)
            if o[1] and o[3] == 1:
                checked += 1
                assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
                nt.assert_allclose(q[i], o[0], atol=1e-6)
        assert checked >= 25 and ok.mean() >= 0.9
        # random starts, C-solver flavour: the advisor's scenario (57 % with the rank-deficient 6 x 6 solve, 100 % with pinv)
        qc, okc, itc, sec, Ec = emu.ik(ets, Tep, method=method, flavour=0, seed=4, k=0.0)
        assert okc.mean() >= 0.95
        good = okc == 1
        err = np.abs(oracle.fkine(ch, qc[good]) - Tep[good]).reshape(good.sum(), -1).max(axis=1)
        assert err.max() < 2e-3
        if ref_harness.available():
            ref = ref_harness.RefETS(ch)
            fn = ref.ik_GN if method == "gn" else ref.ik_NR
            q1, ok1, it1, se1, E1 = emu.ik(ets, Tep, q0=q0, method=method, flavour=0, seed=3, k=0.0)
            hit = 0
            for i in range(N):
                r = fn(Tep[i], q0=q0[i])
                if r[1] == 1 and r[3] == 1:
                    hit += 1
                    assert (r[1], r[2], r[3]) == (ok1[i], it1[i], se1[i])
                    nt.assert_allclose(q1[i], r[0], atol=1e-6)
            assert hit >= 25
    # a position-only mask leaves 3 rows <= n: the row form again (weights drop out of the minimum-norm solution)
    qm, okm, itm, sem, Em = emu.ik(ets, Tep, q0=q0, method="gn", flavour=0, seed=3, k=0.0, mask=[1, 1, 1, 0, 0, 0])
    assert okm.mean() >= 0.9


@pytest.mark.parametrize("robot,kw", [("AL5D", dict(kj=0.05, ks=1.0)), ("px100", dict(kj=0.01, ks=2.0)), ("UR5", dict(kj=0.01, ks=1.0)),
                                      ("Fetch", dict(kj=0.02, ks=1.0, km=500.0)), ("KinovaGen3", dict(kj=1.0, ks=1.0)),
                                      ("UR5", dict(kj=0.01, ks=1.0, kq=1.0, ps=0.0, pi=0.6)), ("KinovaGen3", dict(kj=0.05, ks=1.0, kq=0.5, km=200.0, ps=0.05, pi=0.5))])
def test_ik_qp_joint_counts_against_the_kkt_restatement(robot, kw):
    """IK_QP (robot/IK.py:1222-1520) on chains of 4, 5, 6, 9 and 10 joints: the device's step (closed form: a minimum-norm step
    damped by kj sum|e| / ks, plus the manipulability term where km > 0; a primal-dual active set where kq > 0 adds velocity-damper
    rows) inside the Python solver's loop against the NumPy restatement that builds the reference's Q, c, Aeq, beq, Ain, bin and
    solves the programme by enumerating active sets (oracle/qp.py: a different method on purpose).  Every target, every search: the QP step is
    damped, so nothing is chaotic here.  (Panda: pinned on the reference's own IK_QP code in test_03_python_ik_pins.py.)"""
    from rtbhip import urdf
    from helpers import chain_from_ets
    ets = urdf.load(robot).ets()
    ets.qlim = np.clip(ets.qlim, -np.pi, np.pi)
    ch = chain_from_ets(ets)
    n = ets.n
    rng = np.random.default_rng(100 + n)
    N = 16
    span = ch.qlim[1] - ch.qlim[0]
    qs = rng.uniform(ch.qlim[0] + 0.1 * span, ch.qlim[1] - 0.1 * span, (N, n))
    Tep = oracle.fkine(ch, qs)
    q0 = np.clip(qs + 0.05 * span * rng.normal(size=qs.shape), ch.qlim[0], ch.qlim[1])
    emu.ik_qp_ks(kw["ks"])
    emu.ik_nullspace(kw.get("kq", 0.0), kw.get("km", 0.0), kw.get("ps", 0.0), kw.get("pi", 0.3))
    try:
        q, ok, it, se, E = emu.ik(ets, Tep, q0=q0, method="qp", flavour=1, seed=6, slimit=4, k=kw["kj"])
        wav = emu.ik(ets, Tep, q0=q0, method="qp", flavour=1, seed=6, slimit=4, k=kw["kj"], waves=2)
    finally:
        emu.ik_qp_ks()
        emu.ik_nullspace()
    for a, b in zip((q, ok, it, se, E), wav):
        nt.assert_array_equal(a, b)
    for i in range(N):
        rs = np.array([q0[i]] + [emu.ik_restart(ets, 6, i, d) for d in range(1, 4)])
        o = oracle.ikine_py(ch, Tep[i], rs, step="qp", slimit=4, **kw)
        assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i]), (robot, i)
        if o[1]:
            nt.assert_allclose(q[i], o[0], atol=1e-7)
    assert ok.mean() >= (0.5 if "km" not in kw else 0.2)


@pytest.mark.parametrize("ilimit,slimit", [(20, 100), (20, 200), (30, 100)])
def test_ik_watchdog_budget_covers_the_pass_latency(ilimit, slimit):
    """64 unreachable targets in one wave, the scheduling pass only every 4th iteration (the production default): every
    search of every slot fails and each finished search waits for its pass.  The replay carries the kernel's watchdog
    counter and returns an error if it would have fired (it did, for ilimit % 4 == 0, before the pass latency entered
    the budget: valid failure results would have been overwritten with NaN markers)."""
    import os
    ets, ch = _panda_limited()
    Tep = oracle.fkine(ch, np.zeros((64, 7)))
    Tep[:, :3, 3] += 5.0
    os.environ["EMU_IK_PASS_MASK"] = "3"
    try:
        q, ok, it, se, E = emu.ik(ets, Tep, ilimit=ilimit, slimit=slimit, seed=1, waves=1)
    finally:
        del os.environ["EMU_IK_PASS_MASK"]
    assert not ok.any() and np.all(se == slimit + 1) and np.all(it == slimit * (ilimit + 1))     # ik.cpp:39,66-68
    assert np.all(np.isfinite(q))


@pytest.mark.parametrize("robot,n", [("Fetch", 10), ("KinovaGen3", 9)])
def test_ik_nine_to_twelve_joint_chains(robot, n):
    """IK on chains of 9..12 joints (URDF Fetch: torso + 7-joint arm + gripper path, Kinova Gen3 + finger): the
    same search functions at a larger compile-time joint count; must equal the oracle's sequential loops."""
    from rtbhip import urdf
    from helpers import chain_from_ets
    ets = urdf.load(robot).ets()
    assert ets.n == n
    ets.qlim = np.clip(ets.qlim, -np.pi, np.pi)
    ch = chain_from_ets(ets)
    rng = np.random.default_rng(n)
    N = 12
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, n)))
    q, ok, it, se, E = emu.ik(ets, Tep, seed=5, slimit=30)
    for i in range(N):
        rs = np.array([emu.ik_restart(ets, 5, i, d) for d in range(31)])
        o = oracle.ik_lm(ch, Tep[i], restarts=rs, slimit=30)
        assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
        nt.assert_allclose(q[i], o[0], atol=1e-6)
    b = emu.ik(ets, Tep, seed=5, slimit=30, waves=2)
    for x, y in zip((q, ok, it, se, E), b):
        nt.assert_array_equal(x, y)
    assert ok.mean() >= 0.5


def test_ik_thirteen_to_sixteen_joint_chains():
    """The IK search functions at compile-time joint counts 13..16 (on the device their normal equations spill to scratch):
    sequential specification and wave scheduler against the oracle's loops on a 14- and a 16-joint arm."""
    from helpers import chain_from_ets
    for n in (14, 16):
        arm = rtbhip.DHRobot([rtbhip.RevoluteDH(a=0.04 + 0.01 * (k % 4), d=0.05, alpha=[0.0, np.pi / 2, -np.pi / 2][k % 3]) for k in range(n)]).ets()
        ch = chain_from_ets(arm)
        rng = np.random.default_rng(n)
        N = 8
        Tep = oracle.fkine(ch, rng.uniform(-2, 2, (N, n)))
        q, ok, it, se, E = emu.ik(arm, Tep, seed=9, slimit=6)
        for i in range(N):
            rs = np.array([emu.ik_restart(arm, 9, i, d) for d in range(7)])
            o = oracle.ik_lm(ch, Tep[i], restarts=rs, slimit=6)
            assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i])
            nt.assert_allclose(q[i], o[0], atol=1e-6)
        b = emu.ik(arm, Tep, seed=9, slimit=6, waves=2)
        for x, y in zip((q, ok, it, se, E), b):
            nt.assert_array_equal(x, y)
        assert ok.mean() >= 0.5


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("slimit,with_q0", [(100, False), (30, True), (23, False), (22, False), (7, False), (6, True)])
def test_ik_phased_schedule_equals_sequential_searches(flavour, slimit, with_q0):
    """The phased schedule (rtbhip_tune "ik_phased": first 6 searches of every target, the next 16 of the unresolved ones,
    the rest split into up to 8 work items per target, merged in search order) must report exactly what the sequential loops
    report -- whatever the split points, for unreachable targets, joint-limit rejections and a supplied q0."""
    import os
    ets, ch = _panda_limited()
    rng = np.random.default_rng(slimit)
    N = 150
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    Tep[::13, :3, 3] += 2.5
    q0 = rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)) if with_q0 else None
    a = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=21)
    os.environ["EMU_IK_PHASED"] = "1"
    os.environ["EMU_IK_PASS_MASK"] = "3"
    try:
        b = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=21, waves=3)
    finally:
        del os.environ["EMU_IK_PHASED"], os.environ["EMU_IK_PASS_MASK"]
    for x, y in zip(a, b):
        nt.assert_array_equal(x, y)
    assert a[1].sum() < N


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("N,waves,slimit,with_q0,l0,length", [(300, 6, 100, False, 4, 8), (64, 3, 100, False, 4, 16), (130, 4, 17, True, 4, 8), (5, 4, 100, False, 1, 1),
                                                              (700, 2, 40, False, 8, 5), (200, 5, 4, False, 4, 8), (200, 5, 5, True, 4, 8), (90, 7, 100, False, 3, 97)])
def test_ik_flat_schedule_equals_sequential_searches(flavour, N, waves, slimit, with_q0, l0, length):
    """The flat schedule (ik_device.h: every target's search range cut into chunks, the (target, chunk) items numbered chunk-major and
    drawn from the one device-wide counter by whichever wave has idle lanes; later chunks of a target that has succeeded are skipped or
    dropped; rows merged chunk by chunk) must report exactly what the sequential loops report -- for every cut of the range (a range
    no longer than the first chunk falls back to the plain schedule), unreachable targets, joint-limit rejections and a supplied q0."""
    import os
    ets, ch = _panda_limited()
    rng = np.random.default_rng(N + waves + slimit)
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    Tep[::9, :3, 3] += 2.5
    q0 = rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)) if with_q0 else None
    a = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=41)
    os.environ.update(EMU_IK_FLAT="1", EMU_IK_PASS_MASK="3", EMU_IK_FLAT_L0=str(l0), EMU_IK_FLAT_LEN=str(length))
    try:
        st = [0, 0, 0, 0]
        b = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=41, waves=waves, stats=st)
    finally:
        for k in ("EMU_IK_FLAT", "EMU_IK_PASS_MASK", "EMU_IK_FLAT_L0", "EMU_IK_FLAT_LEN"):
            del os.environ[k]
    for x, y in zip(a, b):
        nt.assert_array_equal(x, y)
    assert a[1].sum() < N and st[0] > 0
    # the regime in which first chunks (almost) never fail -- the ik_benchmark-notebook setting, k = 0.1 without joint-limit rejection:
    # waves without a failed first chunk of their own do not draw later chunks and leave; drawn-but-unstarted numbers must still start
    ok_T = np.delete(Tep, np.s_[::9], axis=0)
    a2 = emu.ik(ets, ok_T, slimit=slimit, flavour=flavour, seed=41, joint_limits=False, k=0.1)
    os.environ.update(EMU_IK_FLAT="1", EMU_IK_PASS_MASK="3", EMU_IK_FLAT_L0=str(l0), EMU_IK_FLAT_LEN=str(length))
    try:
        b2 = emu.ik(ets, ok_T, slimit=slimit, flavour=flavour, seed=41, joint_limits=False, k=0.1, waves=waves)
    finally:
        for k in ("EMU_IK_FLAT", "EMU_IK_PASS_MASK", "EMU_IK_FLAT_L0", "EMU_IK_FLAT_LEN"):
            del os.environ[k]
    for x, y in zip(a2, b2):
        nt.assert_array_equal(x, y)


@pytest.mark.parametrize("flavour", [0, 1])
@pytest.mark.parametrize("N,waves,slimit,with_q0", [(300, 6, 100, False), (64, 3, 100, False), (130, 4, 17, True), (5, 4, 100, False), (700, 2, 40, False)])
def test_ik_cross_wave_sharing_equals_sequential_searches(flavour, N, waves, slimit, with_q0):
    """Cross-wave sharing (ik_device.h: a wave out of work takes a ticket and is handed the unstarted part of another wave's
    search range as a new work item; a target's rows are chained in search order and merged at the end) must report exactly what
    the sequential loops report; the replay also checks that every wave ends on an unserved ticket, i.e. every appended item
    was handed to somebody."""
    import os
    ets, ch = _panda_limited()
    rng = np.random.default_rng(N + waves)
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    Tep[::9, :3, 3] += 2.5
    q0 = rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)) if with_q0 else None
    a = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=31)
    os.environ["EMU_IK_SHARE"] = "1"
    os.environ["EMU_IK_PASS_MASK"] = "3"
    try:
        for after in ("0", "3"):                 # ranges cut at once / only from targets with three failed searches (the default)
            os.environ["EMU_IK_DONATE_AFTER"] = after
            b = emu.ik(ets, Tep, q0=q0, slimit=slimit, flavour=flavour, seed=31, waves=waves)
            for x, y in zip(a, b):
                nt.assert_array_equal(x, y)
    finally:
        del os.environ["EMU_IK_SHARE"], os.environ["EMU_IK_PASS_MASK"], os.environ["EMU_IK_DONATE_AFTER"]
    assert a[1].sum() < N


def test_ik_row_block_with_target_base_equals_the_whole_batch():
    """rtbhip_ik_target_base: rows [b, b + c) of a batch solved on their own, with the restart generator keyed from b, give exactly
    what the whole batch gives for those rows (multi-search targets included) -- sharded IK does not depend on the split."""
    ets, ch = _panda_limited()
    rng = np.random.default_rng(21)
    N = 60
    Tep = oracle.fkine(ch, rng.uniform(ch.qlim[0], ch.qlim[1], (N, 7)))
    Tep[::11, :3, 3] += 2.5
    for flavour in (0, 1):
        whole = emu.ik(ets, Tep, seed=12, slimit=25, flavour=flavour)
        assert whole[3].max() > 3                                     # restarts matter in this batch
        for b, c in ((0, 17), (17, 30), (47, 13)):
            emu.ik_target_base(b)
            try:
                part = emu.ik(ets, Tep[b:b + c], seed=12, slimit=25, flavour=flavour, waves=2)
            finally:
                emu.ik_target_base(0)
            for x, y in zip(whole, part):
                nt.assert_array_equal(x[b:b + c], y)
        wrong = emu.ik(ets, Tep[17:47], seed=12, slimit=25, flavour=flavour)
        assert not np.array_equal(wrong[0], whole[0][17:47])          # without the base the block is a different problem


def test_xcd_tile_mapping_is_a_bijection_with_contiguous_eighths():
    """xcd_tile_of (trig.h): workgroup ids b = 8 i + x (XCD x) -> tiles; a permutation of [0, g) for every grid size, each
    XCD's tiles one contiguous block, visited in increasing order."""
    import ctypes as C
    lib = emu.lib()
    lib.emu_xcd_tile.restype = C.c_uint
    lib.emu_xcd_tile.argtypes = [C.c_uint, C.c_uint]
    for g in list(range(1, 70)) + [255, 256, 257, 15625, 15632]:
        tiles = np.array([lib.emu_xcd_tile(g, b) for b in range(g)])
        assert sorted(tiles.tolist()) == list(range(g)), g
        for x in range(min(8, g)):
            mine = tiles[x::8]
            assert np.all(np.diff(mine) == 1), (g, x)


def test_ik_schedule_study_tool_runs_and_its_floor_is_a_floor():
    """tests/tools/ik_schedule_study.py (the CPU study behind profiles/r03_ab_ik_schedule_floor.txt) at a small size: the idealised pool can
    never finish before the useful work alone would, never runs less than the useful work, and the replay's useful work is the
    sequential specification's."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ik_schedule_study", os.path.join(os.path.dirname(__file__), "tools", "ik_schedule_study.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    N, lanes = 3000, 64 * 60
    ets, Tep = tool.targets(N)
    it = emu.ik(ets, Tep, seed=2)[2]
    tau, useful, discarded = tool.floor(N, lanes, 4)
    assert tau * lanes >= useful + discarded >= useful > 0
    assert abs(useful - int(it.sum())) <= 0.02 * it.sum()          # (the model rounds a target's search length)
    assert tool.replay(N, 60) == int(it.sum())
    assert "EMU_IK_FLAT" not in os.environ and "EMU_IK_PASS_MASK" not in os.environ

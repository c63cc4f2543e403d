"""SURVEY rows a7 / a9 pinned on the reference itself: tests/golden/ref_python_ik.npz holds what the reference's OWN
Python solvers (robot/IK.py IK_LM / IK_NR / IK_GN / IK_QP `.solve`, run unmodified under the stand-in modules of
oracle/ref_python.py; IK_QP with an exact KKT solve standing in for the absent qpsolvers/quadprog, kq = 0) and its compiled
`fknm.Angle_Axis` return.  The start-vector tables of those runs are the device
generator's (rtbhip_ik_restart, a host function), so the device's ikine_* -- whose restarts nobody can hand it --
must reproduce the reference's (q, success, iterations, searches, residual) across MANY searches, not only
first-search cases.

  -m "not gpu": the kernel's search functions replayed on the CPU (tests/emu), sequentially and through the wave scheduler
  -m gpu      : ETS.ikine_LM / ikine_NR / ikine_GN and rtbhip_angle_axis on the device
"""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from oracle import chains, ref_python
from helpers import ref_python_ik, PY_IK_CASES, py_ik_problem, angle_axis_tolerance

PY = ref_python_ik()
SEED = int(PY["ik_seed"])


def _panda_limited():
    ets = rtbhip.models.Panda().ets()
    ets.qlim = chains.PANDA_QLIM
    return ets


def test_fixture_start_tables_are_the_device_generators():
    """Staleness guard: the fixtures were generated for THIS restart generator (keyed by seed, target, draw, joint)."""
    ets = _panda_limited()
    tab = PY["ik_starts"]
    for i in (0, 3, tab.shape[0] - 1):
        for d in (0, 1, 17, tab.shape[1] - 1):
            nt.assert_array_equal(ets.ik_restart(SEED, i, d), tab[i, d])


def _check(key, q, ok, it, se, E):
    prob, first, slimit, step, kw = PY_IK_CASES[key]
    meta, qref, Eref = PY[key + "_meta"], PY[key + "_q"], PY[key + "_E"]
    checked = 0
    for i in range(len(meta)):
        # undamped NR / GN searches from far-away random starts are chaotic: compare what the first search decides
        if step not in ("lm", "qp") and not (meta[i, 0] == 1 and meta[i, 2] == 1):
            continue
        if meta[i, 0] == 0 and not np.all(np.isfinite(qref[i])):
            # a search of the reference blew up numerically (q became NaN: numpy.linalg.inv of J^T W J + k E 1 with E -> 0 on a
            # redundant arm, then "SVD did not converge" one step later, robot/IK.py:320-323).  WHICH iteration overflows is a
            # property of the linear-algebra routine, not of the algorithm: the decision, the search count and the iteration
            # count to within one per search are compared
            assert (int(ok[i]), int(se[i])) == (0, int(meta[i, 2])) and abs(int(it[i]) - int(meta[i, 1])) <= int(meta[i, 2]), (key, i)
            checked += 1
            continue
        assert (int(ok[i]), int(it[i]), int(se[i])) == tuple(meta[i]), (key, i)
        if meta[i, 0]:
            nt.assert_allclose(q[i], qref[i], atol=1e-6)
            assert abs(E[i] - Eref[i]) <= 1e-9 + 1e-6 * abs(Eref[i])
        checked += 1
    assert checked >= (len(meta) if step in ("lm", "qp") else 6), (key, checked)


@pytest.mark.parametrize("key", sorted(PY_IK_CASES))
def test_emu_search_functions_equal_reference_python_solvers(key):
    import emu_harness as emu
    ets = _panda_limited()
    prob, first, slimit, step, kw = PY_IK_CASES[key]
    Tep, tab = py_ik_problem(PY, key)
    kw = dict(kw)
    ns = (kw.pop("kq", 0.0), kw.pop("km", 0.0), kw.pop("ps", 0.0), kw.pop("pi", 0.3))
    args = dict(q0=tab[:, 0] if first else None, slimit=slimit, flavour=1, seed=SEED, method=kw.pop("method", step),
                k=kw.pop("kj") if step == "qp" else kw.pop("k", 0.0), ilimit=kw.pop("ilimit", 30),
                joint_limits=kw.pop("joint_limits", True), mask=kw.pop("mask", None))
    emu.ik_qp_ks(kw.pop("ks", 1.0))
    assert not kw
    emu.ik_nullspace(*ns)
    try:
        seq = emu.ik(ets, Tep, **args)
        wav = emu.ik(ets, Tep, waves=2, **args)
    finally:
        emu.ik_nullspace()
        emu.ik_qp_ks()
    _check(key, *seq)
    for a, b in zip(seq, wav):
        nt.assert_array_equal(a, b)


def test_emu_angle_axis_kernel_body_equals_reference_Angle_Axis():
    """k_angle_axis replayed on the CPU (tile loads through LDS, per-lane error vector, staged flush): every branch of
    ik.cpp:241-286 against the reference's compiled Angle_Axis, ragged tiles, broadcasting of a single pose."""
    import emu_harness as emu
    Te, Tep, ref = PY["aa_Te"], PY["aa_Tep"], PY["aa_e"]
    e = emu.angle_axis(Te, Tep)
    assert np.all(np.abs(e - ref).max(axis=1) <= angle_axis_tolerance(Te, Tep))
    assert np.abs(e - ref)[PY["aa_tag"] <= 2].max() <= 5e-15
    for n in (1, 63, 64, 65):
        nt.assert_array_equal(emu.angle_axis(Te[:n], Tep[:n]), e[:n])
    nt.assert_array_equal(emu.angle_axis(Te[7], Tep[:70]), np.array([emu.angle_axis(Te[7], Tep[i])[0] for i in range(70)]))
    nt.assert_array_equal(emu.angle_axis(Te[:70], Tep[9]), np.array([emu.angle_axis(Te[i], Tep[9])[0] for i in range(70)]))


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(PY_IK_CASES))
def test_gpu_ikine_equals_reference_python_solvers(key):
    ets = _panda_limited()
    prob, first, slimit, step, kw = PY_IK_CASES[key]
    Tep, tab = py_ik_problem(PY, key)
    kw = dict(kw)
    fn = {"lm": ets.ikine_LM, "nr": ets.ikine_NR, "gn": ets.ikine_GN, "qp": ets.ikine_QP}[step]
    if step in ("nr", "gn"):
        kw["pinv"] = True
    sol = fn(Tep, q0=tab[:, 0] if first else None, slimit=slimit, seed=SEED, **kw)
    e = sol.each
    _check(key, sol.q, e["success"], e["iterations"], e["searches"], e["residual"])


@pytest.mark.gpu
def test_gpu_angle_axis_equals_reference_Angle_Axis():
    """rtbhip_angle_axis (batched fknm.Angle_Axis) on the fixture's random / identical / half-turn / near-threshold pairs,
    host and device pointers; broadcasting of one Te against many Tep as tools/p_servo.py uses it."""
    import torch
    Te, Tep, ref = PY["aa_Te"], PY["aa_Tep"], PY["aa_e"]
    e = rtbhip.angle_axis(Te, Tep)
    assert e.shape == ref.shape
    assert np.all(np.abs(e - ref).max(axis=1) <= angle_axis_tolerance(Te, Tep))
    assert np.abs(e - ref)[PY["aa_tag"] <= 2].max() <= 5e-15
    ed = rtbhip.angle_axis(torch.from_numpy(Te).cuda(), torch.from_numpy(Tep).cuda())
    nt.assert_array_equal(ed.cpu().numpy(), e)
    one = rtbhip.angle_axis(Te[0], Tep[0])
    assert one.shape == (6,)
    nt.assert_array_equal(one, e[0])
    nt.assert_array_equal(rtbhip.angle_axis(Te[0], Tep[:5]), np.array([rtbhip.angle_axis(Te[0], Tep[i]) for i in range(5)]))


@pytest.mark.skipif(not ref_python.available(), reason="needs /root/reference (build container)")
def test_fixtures_are_what_the_reference_returns_today():
    """Build container only: re-run two fixture cases through the live reference IK.py and Angle_Axis."""
    ch = chains.panda_ets(with_limits=True)
    duck = ref_python.DuckETS(ch)
    for key in ("lm_chan", "lm_chan_ns"):
        prob, first, slimit, step, kw = PY_IK_CASES[key]
        Tep, tab = py_ik_problem(PY, key)
        for i in (0, 5, 9):
            r = ref_python.solve("IK_LM", duck, Tep[i], tab[i], slimit=slimit, **kw)
            assert (r[1], r[2], r[3]) == tuple(PY[key + "_meta"][i])
            nt.assert_array_equal(r[0], PY[key + "_q"][i])
    for i in (0, 70, 80, 95):
        nt.assert_array_equal(ref_python.angle_axis(PY["aa_Te"][i], PY["aa_Tep"][i]), PY["aa_e"][i])

"""rtbhip.compat.fknm / rtbhip.compat.frne: plug-in modules with the reference extension modules' own function tables
(core/fknm.cpp:23-93, core/frne.c:42-62).

  -m "not gpu": the name tables, handle plumbing, argument / error behaviour that needs no device
  -m gpu      : oracle/ref_harness.py -- written for the reference's compiled modules -- run UNCHANGED with the shims
                substituted, and compared function by function with the reference's own build (oracle/_ref) where that
                is present, with the reference-run fixtures otherwise.
"""
import contextlib
import os
import re

import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
import rtbhip.compat as compat
from oracle import chains, ref_harness as rh
from helpers import ref_outputs, ref_python_ik, tool_base

REF = ref_outputs()
FKNM_NAMES = ["Angle_Axis", "IK_GN_c", "IK_NR_c", "IK_LM_c", "Robot_link_T", "ETS_hessian0", "ETS_hessiane", "ETS_jacobe",
              "ETS_jacob0", "ETS_fkine", "ETS_init", "ET_update", "ET_init", "ET_T"]          # fknm.cpp:23-93 minus r2q
FRNE_NAMES = ["init", "frne", "delete"]                                                          # frne.c:42-62


@contextlib.contextmanager
def harness_on(fknm, frne):
    """ref_harness resolves its two modules through a cache; point it at other modules with the same API."""
    saved = dict(rh._MODS)
    rh._MODS["fknm"], rh._MODS["frne"] = fknm, frne
    try:
        yield
    finally:
        rh._MODS.clear()
        rh._MODS.update(saved)


def test_shim_exports_the_reference_function_tables():
    for n in FKNM_NAMES:
        assert callable(getattr(compat.fknm, n)), n
    for n in FRNE_NAMES:
        assert callable(getattr(compat.frne, n)), n
    src = "/root/reference/src/roboticstoolbox/core/fknm.cpp"
    if os.path.exists(src):                                   # build container: the list above IS the reference's table
        table = re.findall(r'\{"(\w+)",\s*\(PyCFunction\)', open(src).read())
        assert sorted(set(table) - {"r2q"}) == sorted(FKNM_NAMES)
        table = re.findall(r'"(\w+)",\s*\(PyCFunction\)', open(src.replace("fknm.cpp", "frne.c")).read())
        assert sorted(table) == sorted(FRNE_NAMES)


def test_install_and_opt_in_switch(monkeypatch):
    import sys
    monkeypatch.delenv("RTB_BACKEND", raising=False)
    assert compat.auto("rtb_shim_test_pkg") is False and "rtb_shim_test_pkg.fknm" not in sys.modules
    monkeypatch.setenv("RTB_BACKEND", "rtbhip")
    try:
        assert compat.auto("rtb_shim_test_pkg") is True
        from importlib import import_module
        assert import_module("rtb_shim_test_pkg.fknm") is compat.fknm and import_module("rtb_shim_test_pkg.frne") is compat.frne
    finally:
        sys.modules.pop("rtb_shim_test_pkg.fknm", None); sys.modules.pop("rtb_shim_test_pkg.frne", None)


def test_handles_and_error_behaviour_without_a_device():
    f = compat.fknm
    T = np.asfortranarray(np.eye(4)); ql = np.array([-1.0, 1.0])
    et = f.ET_init(0, 1, 0, 0, 2, T, ql)
    ets = f.ETS_init([et], 1, 1)
    assert ets.handle() != 0 and ets.q_width == 1
    h0 = ets.handle()
    assert f.ET_update(et, 0, 1, 1, 0, 0, T, ql) is et          # same element, now a flipped Rx
    assert ets.handle() != h0                                     # the chain was recompiled
    with pytest.raises(TypeError, match="Symbolic value"):       # fknm.cpp:1304-1318
        f.ETS_fkine(ets, np.array([object()]), None, None, 1)
    with pytest.raises(TypeError, match="Symbolic value"):
        f.ET_T(et, 1)                                             # PyFloat_Check (fknm.cpp:1263-1273): an int is refused
    with pytest.raises(TypeError, match="Symbolic value"):
        f.Angle_Axis(np.array([["a"]]), np.eye(4))
    with pytest.raises(TypeError):
        f.ET_init(0, 1, 0, 0, 2, [[1, 0], [0, 1]], ql)            # "O!" wants ndarrays
    with pytest.raises(ValueError):
        f.ETS_jacob0(object(), np.zeros(1), None)                 # not an ETS handle
    sym = f.ET_init(1, 0, 0, 0, 2, T, ql)
    with pytest.raises(TypeError, match="Symbolic value"):
        f.ET_T(sym, None)
    r = compat.frne.init(1, 0, np.zeros(24), [0, 0, 9.81])
    assert compat.frne.delete(r) == 1
    with pytest.raises(ValueError):
        compat.frne.frne(r, [0.0], [0.0], [0.0], [0, 0, 9.81], np.zeros(6))     # deleted
    # the unmodified harness builds its chains on the shim without touching a device
    with harness_on(compat.fknm, compat.frne):
        e = rh.RefETS(chains.panda_ets(with_limits=True))
        assert e.cap.n == 7 and e.cap.handle() != 0
        assert rh.RefRNE(chains.puma560().L24(), 0).cap.n == 6


# ---------------------------------------------------------------------------------------------- GPU
def _both(ch):
    """(RefETS on the shim, RefETS on the reference's own build or None)."""
    with harness_on(compat.fknm, compat.frne):
        shim = rh.RefETS(ch)
    return shim, (rh.RefETS(ch) if rh.available() else None)


@pytest.mark.gpu
def test_harness_on_shim_kinematics_module_for_module():
    tool, base = tool_base()
    rng = np.random.default_rng(5)
    for ch, q in ((chains.panda_ets(with_limits=True), REF["panda_q"]), (chains.Chain(__import__("helpers").mixed_spec(), name="mixed"), REF["mixed_q"])):
        shim, ref = _both(ch)
        Tt, Tb = np.asfortranarray(tool), np.asfortranarray(base)
        # shapes and memory orders of the reference's returns
        one = shim.fknm.ETS_fkine(shim.cap, q[0], None, None, 1)
        assert one.shape == (4, 4) and one.flags.f_contiguous
        J1 = shim.fknm.ETS_jacob0(shim.cap, q[0], None)
        assert J1.shape == (6, ch.n) and J1.flags.f_contiguous
        traj = shim.fkine(q)
        assert traj.shape == (len(q), 4, 4) and traj.flags.c_contiguous
        for qq in (q[0], list(q[0]), q[0][None, :], q[0][:, None]):           # tests/test_ETS.py:359-362
            nt.assert_allclose(shim.jacob0(qq), J1, atol=1e-14)
            nt.assert_allclose(shim.fkine(qq), one, atol=1e-14)
        if ref is not None:
            nt.assert_allclose(traj, ref.fkine(q), atol=1e-10)
            nt.assert_allclose(shim.fkine(q, base=Tb, tool=Tt), ref.fkine(q, base=Tb, tool=Tt), atol=1e-10)
            nt.assert_allclose(shim.fkine(q, base=Tb, tool=Tt, include_base=False), ref.fkine(q, base=Tb, tool=Tt, include_base=False), atol=1e-10)
            nt.assert_allclose(shim.jacob0_batch(q[:16]), ref.jacob0_batch(q[:16]), atol=1e-10)
            nt.assert_allclose(shim.jacobe_batch(q[:16], tool=Tt), ref.jacobe_batch(q[:16], tool=Tt), atol=1e-10)
            for i in range(4):
                nt.assert_allclose(shim.hessian0(q[i]), ref.hessian0(q[i]), atol=1e-10)          # the (ets, q, J, tool) form, J supplied
                Je = ref.jacobe(q[i])
                nt.assert_allclose(shim.fknm.ETS_hessiane(shim.cap, q[i], Je, None), ref.fknm.ETS_hessiane(ref.cap, q[i], Je, None), atol=1e-10)
                nt.assert_allclose(shim.fknm.ETS_hessian0(shim.cap, q[i], None, Tt), ref.fknm.ETS_hessian0(ref.cap, q[i], None, Tt), atol=1e-10)
        else:
            key = "panda" if ch.n == 7 and ch.m > 12 else "mixed"
            nt.assert_allclose(traj, REF[key + "_fkine"], atol=1e-10)
            nt.assert_allclose(shim.jacob0_batch(q), REF[key + "_jacob0"], atol=1e-10)
        # batch extension: the N-row call equals the row-by-row calls
        nt.assert_array_equal(shim.fknm.ETS_jacob0(shim.cap, q[:5], None), shim.jacob0_batch(q[:5]))


@pytest.mark.gpu
def test_harness_on_shim_ik_and_angle_axis():
    ch = chains.panda_ets(with_limits=True)
    shim, ref = _both(ch)
    Tep, q0 = REF["ik_Tep"], REF["ik_q0"]
    for meth, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):
        meta = REF["ik_%s_meta" % meth]
        first = (meta[:, 2] == 1) & (meta[:, 0] == 1)
        for i in np.where(first)[0][:8]:
            q, ok, it, se, E = shim.ik_LM(Tep[i], q0=q0[i], k=k, method=meth)
            assert isinstance(ok, int) and isinstance(it, int) and isinstance(E, float) and q.shape == (7,)
            assert (ok, it, se) == tuple(meta[i])
            nt.assert_allclose(q, REF["ik_%s_q" % meth][i], atol=1e-6)
            if ref is not None:
                r = ref.ik_LM(Tep[i], q0=q0[i], k=k, method=meth)
                assert (r[1], r[2], r[3]) == (ok, it, se)
                nt.assert_allclose(q, r[0], atol=1e-6)
    for fn, key in (("ik_GN", "ik_gn"), ("ik_NR", "ik_nr")):
        meta = REF[key + "_meta"]
        first = (meta[:, 2] == 1) & (meta[:, 0] == 1)
        for i in np.where(first)[0][:6]:
            q, ok, it, se, E = getattr(shim, fn)(Tep[i], q0=q0[i])
            assert (ok, it, se) == tuple(meta[i])
            nt.assert_allclose(q, REF[key + "_q"][i], atol=1e-6)
    # batch extension + the first-letter method dispatch of fknm.cpp:481-495
    qb, okb, itb, seb, Eb = shim.fknm.IK_LM_c(shim.cap, Tep[:6], q0[:6], 30, 100, 1e-6, 1, None, 1.0, "chan")
    hit = 0
    for i in range(6):
        r = shim.ik_LM(Tep[i], q0=q0[i])
        if r[3] == 1 and seb[i] == 1:          # later searches draw restarts keyed by the target's index IN ITS BATCH
            hit += 1
            assert (r[1], r[2], r[3]) == (okb[i], itb[i], seb[i])
            nt.assert_allclose(qb[i], r[0], atol=1e-12)
    assert hit >= 3
    a = shim.fknm.IK_LM_c(shim.cap, Tep[0], q0[0], 30, 100, 1e-6, 1, None, 0.01, "sugi-anything")
    b = shim.fknm.IK_LM_c(shim.cap, Tep[0], q0[0], 30, 100, 1e-6, 1, None, 0.01, "sugihara")
    nt.assert_array_equal(a[0], b[0])
    PY = ref_python_ik()
    e = np.array([shim.fknm.Angle_Axis(x, y) for x, y in zip(PY["aa_Te"][:80], PY["aa_Tep"][:80])])
    assert e.shape == (80, 6)
    nt.assert_allclose(e[PY["aa_tag"][:80] <= 2], PY["aa_e"][:80][PY["aa_tag"][:80] <= 2], atol=5e-15)
    if ref is not None:
        for i in (0, 65, 72):
            nt.assert_allclose(shim.fknm.Angle_Axis(PY["aa_Te"][i], PY["aa_Tep"][i]), ref.fknm.Angle_Axis(PY["aa_Te"][i], PY["aa_Tep"][i]), atol=5e-15)


@pytest.mark.gpu
def test_harness_on_shim_frne_and_scene_graph_calls():
    puma = chains.puma560()
    with harness_on(compat.fknm, compat.frne):
        shim = rh.RefRNE(puma.L24(), 0)
    q, qd, qdd = REF["puma_q"], REF["puma_qd"], REF["puma_qdd"]
    one = shim.frne.frne(shim.cap, q[0], qd[0], qdd[0], -shim.gravity, np.zeros(6))
    assert isinstance(one, list) and len(one) == 6 and all(isinstance(v, float) for v in one)      # frne.c:222-226
    def rel(a, b):
        return np.abs(a - b).max() / max(1.0, np.abs(b).max())
    assert rel(shim.rne(q, qd, qdd), REF["puma_rne"]) <= 1e-9
    assert rel(shim.rne(q, qd, qdd, fext=[1, 2, 3, 1, 2, 3]), REF["puma_rne_fext"]) <= 1e-9
    assert rel(shim.rne(q, qd, qdd, gravity=[1.5, -2.0, -9.0]), REF["puma_rne_gx"]) <= 1e-9
    batch = shim.frne.frne(shim.cap, q, qd, qdd, -shim.gravity, np.zeros(6))                       # batch extension
    assert batch.shape == q.shape and rel(batch, REF["puma_rne"]) <= 1e-9
    if rh.available():
        ref = rh.RefRNE(puma.L24(), 0)
        assert rel(shim.rne(q[:8], qd[:8], qdd[:8]), ref.rne(q[:8], qd[:8], qdd[:8])) <= 1e-9
    shim.delete()
    # ET_T and Robot_link_T (scene-graph refresh) through the same kernels
    f = compat.fknm
    T0 = np.asfortranarray(np.eye(4)); ql = np.array([-3.0, 3.0])
    for axis in range(6):
        for flip in (0, 1):
            et = f.ET_init(0, 1, flip, 0, axis, T0, ql)
            A = f.ET_T(et, 0.7)
            assert A.shape == (4, 4) and A.flags.f_contiguous
            names = ["Rx", "Ry", "Rz", "tx", "ty", "tz"]
            nt.assert_allclose(A, chains.elementary(names[axis], -0.7 if flip else 0.7), atol=1e-15)
    C0 = np.asfortranarray(chains.elementary("Rz", 0.3) @ chains.elementary("tx", 0.2))
    nt.assert_allclose(f.ET_T(f.ET_init(0, 0, 0, 0, 0, C0, ql), None), C0, atol=1e-15)
    ch = chains.panda_ets()
    with harness_on(compat.fknm, compat.frne):
        e = rh.RefETS(ch)
    Tl = [np.asfortranarray(np.zeros((4, 4))), np.zeros((4, 4))]
    qv = REF["panda_q"][3]
    assert f.Robot_link_T([e.cap, e.cap], Tl, qv.copy(), None) is None
    for T in Tl:
        nt.assert_allclose(T, REF["panda_fkine"][3], atol=1e-10)
    f.Robot_link_T([e.cap], Tl[:1], np.zeros(7), REF["panda_q"][4])
    nt.assert_allclose(Tl[0], REF["panda_fkine"][4], atol=1e-10)

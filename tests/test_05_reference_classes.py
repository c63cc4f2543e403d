"""The reference's OWN Python classes -- robot/ET.py `ET`, robot/ETS.py `ETS`, robot/IK.py behind `ETS.ikine_LM` -- executed
unmodified (oracle/ref_classes.py) and bound

  * to the reference's compiled `fknm` (oracle/_ref): validates the harness itself on the reference's golden literals, and
  * (-m gpu) to `rtbhip.compat.fknm`, the plug-in shim over librtbhip.so: the same class objects and method bodies now run on
    the MI355X, and every array they return is compared with what the same classes return on the reference's extension.

This is the drop-in boundary (SURVEY section 8 row b) exercised where a user of the reference stands: `ETS.eval(q (N,7))`,
`.fkine`, `.jacob0`, `.jacobe`, `.hessian0`, `.hessiane`, `.ik_LM`, `.ikine_LM`, and the symbolic fall-back whose control flow
hangs on the extension raising TypeError (robot/ETS.py:1075-1078, :1196-1199).
"""
import numpy as np
import numpy.testing as nt
import pytest

from helpers import literals
from oracle import ref_classes, ref_harness

pytestmark = pytest.mark.skipif(not (ref_classes.available() and ref_harness.available()),
                                reason="needs oracle/_ref (the reference's compiled extension and byte-compiled classes)")

Q1 = np.array([1.4, 0.2, 1.8, 0.7, 0.1, 3.1, 2.9])


def test_reference_classes_on_reference_extension_reproduce_the_goldens():
    L = literals()
    ns = ref_classes.load_reference()
    panda = ref_classes.panda(ns)
    assert (panda.n, panda.m) == (7, 22) and type(panda).__module__ == "roboticstoolbox.robot.ETS"
    nt.assert_allclose(panda.eval(Q1), L["G1_panda_fkine"], atol=1e-6)              # tests/test_Robot.py:18-33
    for q in (Q1, list(Q1), Q1.reshape(1, 7), Q1.reshape(7, 1)):                     # tests/test_ETS.py:359-362
        nt.assert_allclose(panda.jacob0(q), L["G2_panda_jacob0"], atol=1e-6)
    T = panda.eval(Q1)
    R6 = np.zeros((6, 6)); R6[:3, :3] = T[:3, :3].T; R6[3:, 3:] = T[:3, :3].T
    nt.assert_allclose(panda.jacobe(Q1), R6 @ panda.jacob0(Q1), atol=1e-12)         # G3, tests/test_ETS.py:365-398
    assert len(panda.fkine(np.tile(Q1, (5, 1)))) == 5                                # ETS.fkine wraps the rows in SE3


def _symbolic_fallback(ns, ref_panda):
    """ETS.eval / ETS.jacob0 on sympy symbols: the extension's TypeError("Symbolic value") sends the call to the class's own
    Python arithmetic (robot/ETS.py:1075-1141, :1196-1259); evaluated at Q1 it must be the numeric result."""
    sympy = pytest.importorskip("sympy")
    panda = ref_classes.panda(ns)
    qs = sympy.symbols("q:7")
    T = panda.eval(qs)
    J = panda.jacob0(qs)
    assert T.dtype == object and T.shape == (4, 4) and J.dtype == object and J.shape == (6, 7)
    fT, fJ = sympy.lambdify(qs, sympy.Matrix(T)), sympy.lambdify(qs, sympy.Matrix(J))
    nt.assert_allclose(np.array(fT(*Q1), dtype=float), ref_panda.eval(Q1), atol=1e-12)
    nt.assert_allclose(np.array(fJ(*Q1), dtype=float), ref_panda.jacob0(Q1), atol=1e-12)


def test_symbolic_fallback_on_reference_extension():
    ns = ref_classes.load_reference()
    _symbolic_fallback(ns, ref_classes.panda(ns))


def test_symbolic_fallback_through_the_shim_needs_no_gpu():
    """The shim raises the reference's TypeError before it touches the device: the fall-back control flow works on any host."""
    import rtbhip.compat
    ns = ref_classes.load(rtbhip.compat.fknm, "rtbhip")
    assert ns.fknm is rtbhip.compat.fknm and ns.ETS is not ref_classes.load_reference().ETS
    ref = ref_classes.load_reference()
    _symbolic_fallback(ns, ref_classes.panda(ref))


def test_byte_compiled_classes_are_what_the_gpu_box_loads(monkeypatch):
    """oracle/_ref/pyref/*.pyc (make -f oracle/Makefile refpy) load and behave as the .py files do."""
    import os
    if not all(os.path.exists(ref_classes._pyc(f)) for f in ref_classes.FILES):
        pytest.skip("oracle/_ref/pyref not built")
    monkeypatch.setattr(ref_classes, "REF_PKG", "/nonexistent")
    ns = ref_classes.load(ref_harness._load("fknm"), "ref-pyc")
    assert ns.ETS.__module__ == "roboticstoolbox.robot.ETS"
    ref = ref_classes.load_reference()
    Q = np.random.default_rng(0).uniform(-3, 3, (9, 7))
    nt.assert_array_equal(ref_classes.panda(ns).eval(Q), ref_classes.panda(ref).eval(Q))


# ------------------------------------------------------------------------------------------------ on the device
@pytest.mark.gpu
def test_gpu_reference_classes_on_the_shim():
    import rtbhip.compat
    gpu = ref_classes.load(rtbhip.compat.fknm, "rtbhip")
    ref = ref_classes.load_reference()
    pg, pr = ref_classes.panda(gpu), ref_classes.panda(ref)
    rng = np.random.default_rng(12)
    Q = rng.uniform(-np.pi, np.pi, (257, 7))
    Tg, Tr = pg.eval(Q), pr.eval(Q)
    assert Tg.shape == (257, 4, 4) and Tg.flags.c_contiguous
    nt.assert_allclose(Tg, Tr, atol=1e-10)
    one = pg.eval(Q1)
    assert one.shape == (4, 4) and one.flags.f_contiguous                           # fknm.cpp:1002-1005: Fortran order for one q
    nt.assert_allclose(one, pr.eval(Q1), atol=1e-10)
    base = pr.eval(Q[0]); tool = pr.eval(Q[1])
    nt.assert_allclose(pg.eval(Q[:33], base=base, tool=tool), pr.eval(Q[:33], base=base, tool=tool), atol=1e-10)
    nt.assert_allclose(pg.eval(Q[:33], base=base, tool=tool, include_base=False), pr.eval(Q[:33], base=base, tool=tool, include_base=False), atol=1e-10)
    nt.assert_allclose(pg.eval(Q[:33], base=gpu.SE3(base), tool=gpu.SE3(tool)), pr.eval(Q[:33], base=base, tool=tool), atol=1e-10)   # the shim takes `.A`
    fk = pg.fkine(Q[:5])
    assert len(fk) == 5 and isinstance(fk, gpu.SE3)
    nt.assert_allclose(fk.A, Tr[:5], atol=1e-10)
    for q in (Q1, list(Q1), Q1.reshape(1, 7), Q1.reshape(7, 1), Q[3]):
        nt.assert_allclose(pg.jacob0(q), pr.jacob0(q), atol=1e-10)
        nt.assert_allclose(pg.jacobe(q), pr.jacobe(q), atol=1e-10)
        nt.assert_allclose(pg.jacob0(q, tool=tool), pr.jacob0(q, tool=tool), atol=1e-10)
    nt.assert_allclose(pg.hessian0(Q1), pr.hessian0(Q1), atol=1e-10)
    nt.assert_allclose(pg.hessiane(Q1), pr.hessiane(Q1), atol=1e-10)
    nt.assert_allclose(pg.hessian0(J0=pr.jacob0(Q1)), pr.hessian0(J0=pr.jacob0(Q1)), atol=1e-10)
    nt.assert_allclose(pg.hessian0(Q1, tool=tool), pr.hessian0(Q1, tool=tool), atol=1e-10)
    L = literals()
    nt.assert_allclose(pg.eval(Q1), L["G1_panda_fkine"], atol=1e-6)
    nt.assert_allclose(pg.jacob0(Q1), L["G2_panda_jacob0"], atol=1e-6)
    _symbolic_fallback(gpu, pr)


@pytest.mark.gpu
def test_gpu_reference_classes_inverse_kinematics_on_the_shim():
    import rtbhip.compat
    gpu = ref_classes.load(rtbhip.compat.fknm, "rtbhip")
    ref = ref_classes.load_reference()
    pg, pr = ref_classes.panda(gpu), ref_classes.panda(ref)
    qr = np.array([0, -0.3, 0, -2.2, 0, 2, np.pi / 4])
    Tep = pr.eval(qr)
    q0 = qr + 0.2
    # ETS.ik_LM -> IK_LM_c, first search from q0: no random restart is involved, so the solutions must agree
    for method, k in (("chan", 1.0), ("wampler", 0.01), ("sugihara", 0.01)):           # tests/test_IK.py:632-708 (G10)
        g = pg.ik_LM(Tep, q0=q0, method=method, k=k, tol=1e-6)
        r = pr.ik_LM(Tep, q0=q0, method=method, k=k, tol=1e-6)
        assert g[1] == 1 and r[1] == 1 and g[3] == r[3] == 1
        assert g[2] == r[2]
        nt.assert_allclose(g[0], r[0], atol=1e-6)
        assert g[4] < 1e-6
        e = ref.p_servo.angle_axis(pr.eval(g[0]), Tep)
        assert 0.5 * float(e @ e) < 1e-5                                              # tests/test_IK.py:15 criterion
    g = pg.ik_GN(Tep, q0=q0)
    r = pr.ik_GN(Tep, q0=q0)
    assert g[1] == r[1] == 1
    nt.assert_allclose(g[0], r[0], atol=1e-6)
    # ETS.ikine_LM -> the reference's Python IK_LM, every eval / jacob0 of its loop served by the device, one call each
    sg = pg.ikine_LM(gpu.SE3(Tep), q0=q0, seed=0)
    sr = pr.ikine_LM(ref.SE3(Tep), q0=q0, seed=0)
    assert sg.success and sr.success and sg.iterations == sr.iterations and sg.searches == sr.searches
    nt.assert_allclose(sg.q, sr.q, atol=1e-8)
    sg = pg.ikine_LM(gpu.SE3(Tep), seed=3, joint_limits=False)                        # random restarts: numpy's seeded generator in both
    sr = pr.ikine_LM(ref.SE3(Tep), seed=3, joint_limits=False)
    assert sg.success == sr.success and sg.searches == sr.searches and sg.iterations == sr.iterations
    nt.assert_allclose(sg.q, sr.q, atol=1e-6)

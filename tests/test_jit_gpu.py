"""Run-time instantiation of the structure-signature kernels on the device (csrc/jit.cpp; the part without a GPU: tests/test_jit_cpu.py).

A robot whose structure matches no instantiation built into the library gets its own -- the same hand-written kernel sources compiled by hipRTC
with the robot's structure as template arguments.  The reference has ONE general path for every robot (core/methods.cpp:318-352,
core/ik.cpp:19-75, core/ne.c:62-493, robot/Robot.py:1704-1903), so what must hold is: the run-time kernel returns what the general kernel
returns -- bit for bit, by construction of the structured forms (kin_device.h: dotk; rne_device.h; tree_device.h) -- and what the oracle returns
to the usual tolerance; until the code object is there the general kernel serves and nobody can tell from the numbers."""
import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from rtbhip import jit, urdf
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _modes():
    yield
    for k, v in (("jit", 1), ("rne_sig", 1), ("tree_sig", 1), ("ik_sig", 1), ("sig_builtin", 1), ("diff_sig", 1)):
        rtbhip.tune(k, v)


def need_rtc():
    if not jit.stats()["available"]:
        pytest.skip("libhiprtc.so is not on this box: the general kernels serve (covered by every other test)")


def perturbed_panda(dalpha=0.01):
    p = rtbhip.models.DH.Panda()
    links = list(p.links)
    k = links[3]
    links[3] = rtbhip.RevoluteMDH(a=k.a, d=k.d, alpha=k.alpha + dalpha, m=k.m, r=k.r, I=k.I, G=1)
    return rtbhip.DHRobot(links, name="panda-perturbed")


def _dh_calls(rob, q, qd, qdd, g):
    return {"rne": rob.rne(q, qd, qdd, gravity=g), "gravload": rob.gravload(q, gravity=g), "itorque": rob.itorque(q, qdd),
            "inertia": rob.inertia(q), "coriolis": rob.coriolis(q, qd), "accel": rob.accel(q, qd, qdd, gravity=g)}


def test_dh_robot_without_builtin_instantiation():
    need_rtc()
    rob = perturbed_panda()
    assert len(jit.names(rob)[0]) == 5
    rng = np.random.default_rng(5)
    N = 3001
    q, qd, qdd = rng.uniform(-3, 3, (N, 7)), rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
    g = np.array([0.2, -0.1, 9.81])
    rtbhip.tune("jit", 2)                     # a launch waits for its instantiation
    s0 = jit.stats()
    fast = {k: np.asarray(v) for k, v in _dh_calls(rob, q, qd, qdd, g).items()}
    s1 = jit.stats()
    assert s1["launches"] - s0["launches"] >= 6 and s1["failed"] == s0["failed"], s1
    assert (s1["compiled"] + s1["disk_hits"]) - (s0["compiled"] + s0["disk_hits"]) >= 5
    rtbhip.tune("rne_sig", 0)                 # the general kernels
    general = {k: np.asarray(v) for k, v in _dh_calls(rob, q, qd, qdd, g).items()}
    assert jit.stats()["launches"] == s1["launches"]
    for k in fast:
        nt.assert_array_equal(fast[k], general[k], err_msg=k)       # the same bits
    # and the oracle (core/ne.c restated)
    ref = oracle.rne_dh(rob.L24(), 1, q[:400], qd[:400], qdd[:400], -g)
    assert np.abs(fast["rne"][:400] - ref).max() / max(1.0, np.abs(ref).max()) <= 1e-9


def test_asynchronous_mode_is_invisible_in_the_results():
    need_rtc()
    rob = perturbed_panda(0.02)               # another structure? no: alpha's class is the same -- another TABLE, the same signature; a cache hit at most
    rob2 = perturbed_panda(0.02)
    rng = np.random.default_rng(6)
    q, qd, qdd = rng.uniform(-3, 3, (2000, 7)), rng.normal(size=(2000, 7)), rng.normal(size=(2000, 7))
    rtbhip.tune("jit", 1)
    first = np.asarray(rob.rne(q, qd, qdd))   # general or run-time kernel, whichever is there
    assert jit.wait(120)
    second = np.asarray(rob.rne(q, qd, qdd))
    third = np.asarray(rob2.rne(q, qd, qdd))
    nt.assert_array_equal(first, second)
    nt.assert_array_equal(first, third)
    st = jit.stats()
    assert st["pending"] == 0 and st["launches"] >= 2


@pytest.mark.parametrize("name,end", [("LBR", None), ("Puma560", None), ("px100", None), ("Mico", None)])
def test_ik_of_robots_without_builtin_instantiation(name, end):
    need_rtc()
    e = urdf.load(name).ets(end=end)
    exprs = jit.names(e)[0]
    assert len(exprs) == 2, (name, exprs)
    lim = np.clip(e.qlim, -2.8, 2.8)
    T = np.asarray(e.eval(np.random.default_rng(11).uniform(lim[0], lim[1], (3000, e.n))))
    rtbhip.tune("jit", 2)
    s0 = jit.stats()
    fast = e.ik_LM(T, seed=4)
    s1 = jit.stats()
    assert s1["launches"] > s0["launches"] and s1["failed"] == s0["failed"], s1
    rtbhip.tune("ik_sig", 0)
    general = e.ik_LM(T, seed=4)
    assert jit.stats()["launches"] == s1["launches"]
    # decisions, counts, q and the residual: the same bits, for every robot (the Puma560's q was <= 1e-10 apart until the IK iteration was written
    # contraction-free like the dynamics recursions: ik_device.h, ldl.h)
    for k in (1, 2, 3):
        nt.assert_array_equal(np.asarray(fast[k]), np.asarray(general[k]))
    ok = np.asarray(fast[1]) == 1
    assert ok.mean() > 0.5
    assert np.abs(np.asarray(fast[0])[ok] - np.asarray(general[0])[ok]).max() < 1e-8
    assert np.abs(np.asarray(fast[4])[ok] - np.asarray(general[4])[ok]).max() < 1e-12
    nt.assert_array_equal(np.asarray(fast[0]), np.asarray(general[0]))
    nt.assert_array_equal(np.asarray(fast[4]), np.asarray(general[4]))
    Tq = np.asarray(e.eval(np.asarray(fast[0])[ok]))
    assert np.abs(Tq - T[ok]).max() < 5e-3                           # E = e'e / 2 < 1e-6: the pose error is below 1.5e-3


def _tree_calls(t, q, qd, qdd, g):
    out = {"rne": t.rne(q, qd, qdd, gravity=g), "gravload": t.gravload(q, gravity=g)}
    if t.n <= 20:
        out.update(inertia=t.inertia(q), coriolis=t.coriolis(q, qd), accel=t.accel(q, qd, qdd, gravity=g))
    return out


@pytest.mark.parametrize("name", ["KinovaGen3", "YuMi", "LBR", "Panda", "AL5D", "Puma560"])
def test_link_trees_without_builtin_instantiation(name):
    need_rtc()
    t = urdf.load(name).erobot()
    exprs, pre = jit.names(t)
    assert exprs and "JitTree%d_" % t.n in pre
    rng = np.random.default_rng(7)
    N, n = 1500, t.n
    q, qd, qdd = rng.uniform(-2, 2, (N, n)), rng.normal(size=(N, n)), rng.normal(size=(N, n))
    g = np.array([0.1, 0.2, 9.81])
    rtbhip.tune("jit", 2)
    s0 = jit.stats()
    fast = {k: np.asarray(v) for k, v in _tree_calls(t, q, qd, qdd, g).items()}
    s1 = jit.stats()
    assert s1["launches"] - s0["launches"] >= len(fast) and s1["failed"] == s0["failed"], s1
    rtbhip.tune("tree_sig", 0)
    general = {k: np.asarray(v) for k, v in _tree_calls(t, q, qd, qdd, g).items()}
    assert jit.stats()["launches"] == s1["launches"]
    for k in fast:
        nt.assert_array_equal(fast[k], general[k], err_msg="%s %s" % (name, k))
    # (the general tree kernels against the oracle / the reference: tests/test_erobot_rne.py, test_erobot_dynamics.py -- the evidence transfers)


def test_jit_off_and_stats():
    need_rtc()
    rob = perturbed_panda()
    q = np.random.default_rng(1).uniform(-3, 3, (500, 7))
    rtbhip.tune("jit", 0)
    s0 = jit.stats()
    a = np.asarray(rob.gravload(q))
    assert jit.stats()["launches"] == s0["launches"] and jit.stats()["mode"] == 0
    rtbhip.tune("jit", 2)
    b = np.asarray(rob.gravload(q))
    assert jit.stats()["launches"] == s0["launches"] + 1
    nt.assert_array_equal(a, b)
    st = jit.stats()
    assert st["compile_seconds_max"] < 60 and st["last_error"] == ""


def test_robots_with_a_builtin_instantiation_served_by_their_runtime_one_instead():
    """rtbhip_tune("sig_builtin", 0): the launchers pass over the instantiations built into the library, so the Panda (IK, DH dynamics) and the UR5
    (link tree) take the run-time route like any other robot.  Same sources, same structure words, compiled at another moment: the outputs must be the
    built-in kernels' bits (and a process that never compiles anything -- "jit" = 0 -- falls back to the general kernels, the same bits again)."""
    need_rtc()
    rng = np.random.default_rng(77)
    ets = rtbhip.models.Panda().ets()
    ets.qlim = rtbhip.models.PANDA_QLIM
    Tep = np.asarray(ets.eval(rng.uniform(ets.qlim[0], ets.qlim[1], (1500, 7))))
    dh = rtbhip.models.DH.Panda()
    ur = urdf.load("UR5").erobot()
    N = 1200
    q7, qd7, qdd7 = rng.uniform(-2.5, 2.5, (N, 7)), rng.normal(size=(N, 7)), rng.normal(size=(N, 7))
    q6, qd6, qdd6 = rng.uniform(-2.5, 2.5, (N, 6)), rng.normal(size=(N, 6)), rng.normal(size=(N, 6))

    def everything():
        out = [np.asarray(x) for x in ets.ik_LM(Tep, seed=9)]
        out += [np.asarray(v) for v in _dh_calls(dh, q7, qd7, qdd7, np.array([0.0, 0.0, 9.81])).values()]
        out += [np.asarray(ur.rne(q6, qd6, qdd6)), np.asarray(ur.inertia(q6)), np.asarray(ur.coriolis(q6, qd6)), np.asarray(ur.accel(q6, qd6, qdd6))]
        return out
    assert jit.names(ets)[0] == [] and jit.names(dh)[0] == [] and jit.names(ur)[0] == []       # built in: nothing to compile
    builtin = everything()
    rtbhip.tune("sig_builtin", 0)
    assert len(jit.names(ets)[0]) == 2 and len(jit.names(dh)[0]) == 5 and len(jit.names(ur)[0]) >= 4
    rtbhip.tune("jit", 2)                     # a launch waits for its instantiation
    s0 = jit.stats()
    runtime = everything()
    s1 = jit.stats()
    assert s1["launches"] - s0["launches"] >= 11 and s1["failed"] == s0["failed"], s1
    for a, b in zip(runtime, builtin):
        nt.assert_array_equal(a, b)
    rtbhip.tune("jit", 0)                     # nothing is compiled or looked up: the general kernels
    general = everything()
    assert jit.stats()["launches"] == s1["launches"]
    for a, b in zip(general, builtin):
        nt.assert_array_equal(a, b)


def _diff_calls(e, q, qd):
    out = {"jacob0_dot": e.jacob0_dot(q, qd), "jacobm": e.jacobm(q), "jacobm_trans": e.jacobm(q, axes="trans")}
    for method in ("yoshikawa", "minsingular", "invcondition"):
        for axes in ("all", "trans", "rot"):
            out["manipulability_%s_%s" % (method, axes)] = e.manipulability(q, method=method, axes=axes)
    for rep in ("rpy/xyz", "rpy/zyx", "eul", "exp"):
        out["jacob0_analytical_" + rep] = e.jacob0_analytical(q, rep)
    return {k: np.asarray(v) for k, v in out.items()}


@pytest.mark.parametrize("name", ["Panda", "UR5", "LBR", "Puma560", "px100"])
def test_differential_consumers_by_structure_instantiation(name):
    """k_kin_diff<NJ, MODE, SIG> (csrc/diff_kernel.h): jacob0_dot / manipulability / jacobm / the analytical Jacobians with the walk of THIS robot
    written out, compiled at run time.  The reference computes all of them through one general path (robot/Robot.py:964-1235,
    robot/ETS.py:1687-1819): the instantiation must return the general kernel's bits, and the oracle's numbers."""
    need_rtc()
    e = urdf.load(name).ets()
    rng = np.random.default_rng(21)
    N = 2500
    q, qd = rng.uniform(-3, 3, (N, e.n)), rng.normal(size=(N, e.n))
    rtbhip.tune("jit", 2)
    s0 = jit.stats()
    fast = _diff_calls(e, q, qd)
    s1 = jit.stats()
    assert s1["launches"] - s0["launches"] >= len(fast) and s1["failed"] == s0["failed"], s1
    rtbhip.tune("diff_sig", 0)
    try:
        general = _diff_calls(e, q, qd)
    finally:
        rtbhip.tune("diff_sig", 1)
    assert jit.stats()["launches"] == s1["launches"]
    for k in fast:
        nt.assert_array_equal(fast[k], general[k], err_msg=k)
    from helpers import chain_from_ets
    ch = chain_from_ets(e)
    nt.assert_allclose(fast["jacob0_dot"][:50].reshape(50, 6, e.n), oracle.jacob_dot(ch, q[:50], qd[:50]), atol=1e-10)
    # (a 4-joint arm's six-axis measure is a rounding-level number: its translational one is compared instead)
    axes = "all" if e.n >= 6 else "trans"
    nt.assert_allclose(fast["manipulability_yoshikawa_" + axes][:50], oracle.manipulability(ch, q[:50], [1, 1, 1, 1, 1, 1] if e.n >= 6 else [1, 1, 1, 0, 0, 0]), rtol=1e-9, atol=1e-12)


def test_manifest_of_requested_instantiations(tmp_path):
    """RTBHIP_JIT_MANIFEST=<file>: every instantiation a process asks for is appended as one JSON line {"unit", "expr", "preamble"} -- how the
    ahead-of-time list (jit_aot_manifest.jsonl, compiled by __graft_entry__.build_aot_cache) is gathered on the device."""
    import json, os, subprocess, sys
    need_rtc()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path[:0] = [%r, %r, %r]
import numpy as np, rtbhip
from rtbhip import jit, urdf
from test_jit_gpu import perturbed_panda
rtbhip.tune("jit", 2)
rob = perturbed_panda(0.03)
q = np.random.default_rng(0).uniform(-2, 2, (100, 7))
rob.rne(q, q, q)
t = urdf.load("KinovaGen3").erobot()
q7 = np.zeros((10, t.n))
t.rne(q7, q7, q7)
print("launches", jit.stats()["launches"])
""" % (root, os.path.join(root, "robotics-toolbox-python_amd"), os.path.join(root, "tests"))
    man = tmp_path / "manifest.jsonl"
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RTBHIP_JIT_MANIFEST=str(man)), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and int(out.stdout.split("launches")[1]) >= 2, out.stdout + out.stderr[-1500:]
    rows = [json.loads(l) for l in man.read_text().splitlines()]
    assert all(set(r) == {"unit", "expr", "preamble"} for r in rows)
    assert any(r["unit"] == "rne_kernels.hip" and r["expr"].startswith("rtbhip::k_rne<7, true") for r in rows)
    tree = [r for r in rows if r["unit"] == "tree_kernels.hip"]
    assert tree and "struct" in tree[0]["preamble"]                  # a tree's generated knowledge type travels as the preamble

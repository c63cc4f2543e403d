"""oracle/ref_classes.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Runs the reference's OWN Python classes -- robot/ET.py (`ET`), robot/ETS.py (`ETS`) and, behind `ETS.ik_*` / `ikine_*`,
robot/IK.py -- UNMODIFIED, bound to an `fknm` module of the caller's choice:

    ref = ref_classes.load(ref_harness._load("fknm"), "ref")     # the reference's compiled extension (oracle/_ref): the checker
    gpu = ref_classes.load(rtbhip.compat.fknm, "rtbhip")         # the plug-in shim over librtbhip.so: the thing under test
    panda = lambda ns: ns.ET.tz(0.333) * ns.ET.Rz() * ...        # models/ETS/Panda.py:32-54 with the reference's constructors
    gpu.ETS... .eval(q) == ref.ETS... .eval(q)

This is row (b) of SURVEY section 8 proven at the level the reference's users see: the same class objects, the same method bodies,
only `roboticstoolbox.fknm` resolves elsewhere (what `rtbhip.compat.install()` does in a real installation).

`import roboticstoolbox` itself is impossible here (spatialmath-python is absent); the files are therefore loaded one by one under
stand-in `spatialmath` / `roboticstoolbox` package objects (oracle/sm_standin.py; tools/types.py and tools/p_servo.py are the
reference's own files).  Where /root/reference exists the .py files are executed from where they lie.  On the GPU box it does
not: `make -f oracle/Makefile refpy` (run by __graft_entry__.build() in the build container) byte-compiles the same five files
into oracle/_ref/pyref/*.pyc -- compiled artefacts of the reference next to the compiled fknm / frne, git-ignored, shipped by
gpurun -- and the loader executes those.  No reference source is copied into the repository.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

from . import ref_harness, sm_standin

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PKG = "/root/reference/src/roboticstoolbox"
PYC_DIR = os.path.join(HERE, "_ref", "pyref")
FILES = ["tools/types.py", "tools/p_servo.py", "robot/IK.py", "robot/ET.py", "robot/ETS.py"]
_LOADED = {}


def _pyc(rel):
    return os.path.join(PYC_DIR, rel.replace("/", "_") + "c")


def available():
    return all(os.path.exists(os.path.join(REF_PKG, f)) or os.path.exists(_pyc(f)) for f in FILES)


def compile_pyc():
    """The `refpy` recipe of oracle/Makefile: byte-compile the reference files from where they lie into oracle/_ref/pyref."""
    import py_compile
    os.makedirs(PYC_DIR, exist_ok=True)
    for f in FILES:
        py_compile.compile(os.path.join(REF_PKG, f), cfile=_pyc(f), dfile="roboticstoolbox/" + f, doraise=True)


def _exec(modname, rel):
    src = os.path.join(REF_PKG, rel)
    if os.path.exists(src):
        spec = importlib.util.spec_from_file_location(modname, src)
    elif os.path.exists(_pyc(rel)):
        loader = importlib.machinery.SourcelessFileLoader(modname, _pyc(rel))
        spec = importlib.util.spec_from_loader(modname, loader)
    else:
        raise ImportError("neither %s nor %s exists (run `make -f oracle/Makefile refpy` where /root/reference is)" % (src, _pyc(rel)))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


_NAMES = ("spatialmath", "spatialmath.base", "qpsolvers", "roboticstoolbox", "roboticstoolbox.fknm", "roboticstoolbox.tools",
          "roboticstoolbox.tools.types", "roboticstoolbox.tools.p_servo", "roboticstoolbox.robot", "roboticstoolbox.robot.IK",
          "roboticstoolbox.robot.ET", "roboticstoolbox.robot.ETS")


def load(fknm, tag):
    """A namespace with the reference's `ET`, `ETS`, `IK` module, `SE3` stand-in ... whose `roboticstoolbox.fknm` is `fknm`."""
    if tag in _LOADED:
        return _LOADED[tag]
    saved = {k: sys.modules.get(k) for k in _NAMES}
    try:
        sm, smb = sm_standin.modules()
        sys.modules["spatialmath"], sys.modules["spatialmath.base"] = sm, smb
        qps = types.ModuleType("qpsolvers")

        def solve_qp(*a, **k):
            from . import qp
            return qp.solve_qp(*a, **k)
        qps.solve_qp = solve_qp
        sys.modules["qpsolvers"] = qps
        rtb = types.ModuleType("roboticstoolbox")
        rtb.__path__ = []
        rtb.rtb_get_param = lambda name: {"unicode": True}.get(name)          # tools/params.py:10-11, the one key ETS.__str__ reads
        sys.modules["roboticstoolbox"] = rtb
        sys.modules["roboticstoolbox.fknm"] = fknm
        rtb.fknm = fknm
        tools = types.ModuleType("roboticstoolbox.tools")
        tools.__path__ = []
        sys.modules["roboticstoolbox.tools"] = tools
        rtb.tools = tools
        tools.types = _exec("roboticstoolbox.tools.types", "tools/types.py")
        ps = _exec("roboticstoolbox.tools.p_servo", "tools/p_servo.py")
        rtb.angle_axis, rtb.angle_axis_python, rtb.p_servo = ps.angle_axis, ps.angle_axis_python, ps.p_servo
        robot = types.ModuleType("roboticstoolbox.robot")
        robot.__path__ = []
        sys.modules["roboticstoolbox.robot"] = robot
        ik = _exec("roboticstoolbox.robot.IK", "robot/IK.py")
        et = _exec("roboticstoolbox.robot.ET", "robot/ET.py")
        ets = _exec("roboticstoolbox.robot.ETS", "robot/ETS.py")
        rtb.ET, rtb.ET2, rtb.ETS, rtb.ETS2 = et.ET, et.ET2, ets.ETS, ets.ETS2
        for name in ("IK_LM", "IK_NR", "IK_GN", "IK_QP", "IKSolution", "IKSolver"):
            setattr(rtb, name, getattr(ik, name))
        ns = types.SimpleNamespace(ET=et.ET, ETS=ets.ETS, IK=ik, SE3=sm.SE3, rtb=rtb, fknm=fknm, p_servo=ps, tag=tag)
    finally:
        for k, v in saved.items():          # the classes keep their own references; the stand-ins must not leak to other importers
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _LOADED[tag] = ns
    return ns


def load_reference():
    """The classes on the reference's own compiled extension: the checker."""
    return load(ref_harness._load("fknm"), "ref")


def panda(ns):
    """models/ETS/Panda.py:32-54 spelled with the namespace's (i.e. the reference's) ET constructors, joints numbered explicitly as
    the reference's own tests do when they build the chain without a Robot (tests/test_ETS.py:270-283)."""
    from math import pi
    E = ns.ET
    deg, mm = pi / 180, 1e-3
    tool_offset = (103) * mm
    l0 = E.tz(0.333) * E.Rz(jindex=0)
    l1 = E.Rx(-90 * deg) * E.Rz(jindex=1)
    l2 = E.Rx(90 * deg) * E.tz(0.316) * E.Rz(jindex=2)
    l3 = E.tx(0.0825) * E.Rx(90, "deg") * E.Rz(jindex=3)
    l4 = E.tx(-0.0825) * E.Rx(-90, "deg") * E.tz(0.384) * E.Rz(jindex=4)
    l5 = E.Rx(90, "deg") * E.Rz(jindex=5)
    l6 = E.tx(0.088) * E.Rx(90, "deg") * E.tz(0.107) * E.Rz(jindex=6)
    ee = E.tz(tool_offset) * E.Rz(-pi / 4)
    return l0 + l1 + l2 + l3 + l4 + l5 + l6 + ee

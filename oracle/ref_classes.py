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
# the DH side (load_dh): the classes a `DHRobot` is made of and the two DH models the benchmark configurations name
URDF_FILES = ["tools/urdf/utils.py", "tools/urdf/urdf.py"]
DH_FILES = ["robot/Link.py", "robot/Gripper.py", "robot/RobotProto.py", "robot/DHLink.py", "robot/Dynamics.py", "robot/RobotKinematics.py",
            "robot/BaseRobot.py", "robot/Robot.py", "robot/DHRobot.py", "models/DH/Puma560.py", "models/DH/Panda.py"]
_LOADED = {}


class _Placeholders(types.ModuleType):
    """A module whose every attribute is an empty class: stands for the reference's dependencies that the numeric paths import but
    never call (spatialgeometry shapes, ansitable, the plotting back-ends, the URDF / xacro readers)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


def _pyc(rel):
    return os.path.join(PYC_DIR, rel.replace("/", "_") + "c")


def available():
    return all(os.path.exists(os.path.join(REF_PKG, f)) or os.path.exists(_pyc(f)) for f in FILES)


def dh_available():
    return available() and all(os.path.exists(os.path.join(REF_PKG, f)) or os.path.exists(_pyc(f)) for f in DH_FILES + URDF_FILES)


# the reference's own unit-test files that tests/test_reference_suite.py runs against the GPU backend
REF_TESTS = "/root/reference/tests"
TEST_FILES = ["test_ET.py", "test_ETS.py", "test_jacob.py", "test_IK.py", "test_DHRobot.py", "test_PoERobot.py", "test_ERobot.py", "test_Robot.py",
              "test_tools.py", "test_Link.py", "test_ELink.py"]


def _test_pyc(name):
    return os.path.join(HERE, "_ref", "pytests", name + "c")


def tests_available():
    return all(os.path.exists(os.path.join(REF_TESTS, f)) or os.path.exists(_test_pyc(f)) for f in TEST_FILES)


def load_test_module(name):
    """One of the reference's test files (tests/<name>.py) as a module object: executed from where it lies, or -- on the GPU box -- from
    its byte-compiled copy under oracle/_ref/pytests (make -f oracle/Makefile refpy).  The caller installs the module shims first."""
    fname = name + ".py"
    src = os.path.join(REF_TESTS, fname)
    modname = "reference_tests_" + name
    if os.path.exists(src):
        spec = importlib.util.spec_from_file_location(modname, src)
    elif os.path.exists(_test_pyc(fname)):
        spec = importlib.util.spec_from_loader(modname, importlib.machinery.SourcelessFileLoader(modname, _test_pyc(fname)))
    else:
        raise ImportError("neither %s nor its byte-compiled copy exists" % src)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def compile_pyc():
    """The `refpy` recipe of oracle/Makefile: byte-compile the reference files from where they lie into oracle/_ref/pyref."""
    import py_compile
    os.makedirs(PYC_DIR, exist_ok=True)
    for f in FILES + DH_FILES + URDF_FILES:
        py_compile.compile(os.path.join(REF_PKG, f), cfile=_pyc(f), dfile="roboticstoolbox/" + f, doraise=True)
    os.makedirs(os.path.dirname(_test_pyc("x")), exist_ok=True)
    for f in TEST_FILES:
        py_compile.compile(os.path.join(REF_TESTS, f), cfile=_test_pyc(f), dfile="tests/" + f, doraise=True)


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


class _SceneNode:
    """spatialgeometry.SceneNode as far as robot/Link.py:62,155-160 and robot/BaseRobot.py:73-110 rely on it: a node with a local
    transform `_T`, a parent and a list of children.  (The scene graph itself -- world transforms for a visualiser -- is not restated.)"""

    def __init__(self, T=None, scene_parent=None, scene_children=None):
        import numpy as np
        self._T = np.eye(4) if T is None else np.asarray(T)
        self._wT = np.eye(4)
        self._scene_parent = scene_parent
        self._scene_children = list(scene_children) if scene_children is not None else []

    def _propogate_scene_tree(self):
        pass

    def _update_scene_tree(self, *a, **k):
        pass


class _SceneGroup(_SceneNode):
    def __len__(self): return len(self._scene_children)
    def __iter__(self): return iter(self._scene_children)
    def __getitem__(self, i): return self._scene_children[i]
    def append(self, x): self._scene_children.append(x)


_PLACEHOLDERS = ("spatialgeometry", "ansitable", "roboticstoolbox.backends", "roboticstoolbox.backends.Connector",
                 "roboticstoolbox.backends.PyPlot", "roboticstoolbox.backends.PyPlot.EllipsePlot", "roboticstoolbox.tools.xacro",
                 "roboticstoolbox.tools.URDF", "roboticstoolbox.tools.data", "roboticstoolbox.tools.params")
_NAMES = ("spatialmath", "spatialmath.base", "spatialmath.base.argcheck", "spatialmath.base.symbolic", "qpsolvers", "roboticstoolbox",
          "roboticstoolbox.fknm", "roboticstoolbox.frne", "roboticstoolbox.tools",
          "roboticstoolbox.tools.types", "roboticstoolbox.tools.p_servo", "roboticstoolbox.robot", "roboticstoolbox.robot.IK",
          "roboticstoolbox.robot.ET", "roboticstoolbox.robot.ETS", "roboticstoolbox.models", "roboticstoolbox.models.DH",
          "roboticstoolbox.tools.urdf", "roboticstoolbox.tools.urdf.utils", "roboticstoolbox.tools.urdf.urdf") + _PLACEHOLDERS + tuple(
              "roboticstoolbox." + f[:-3].replace("/", ".") for f in DH_FILES)


def load(fknm, tag, frne=None):
    """A namespace with the reference's `ET`, `ETS`, `IK` module, `SE3` stand-in ... whose `roboticstoolbox.fknm` is `fknm`.
    With an `frne` module as well, the DH side is loaded too (`DHRobot`, `RevoluteDH` ..., `Puma560`, `PandaDH`): see load_dh."""
    if tag in _LOADED:
        return _LOADED[tag]
    saved = {k: sys.modules.get(k) for k in _NAMES}
    try:
        sm, smb = sm_standin.modules()
        sys.modules["spatialmath"], sys.modules["spatialmath.base"] = sm, smb
        sys.modules["spatialmath.base.argcheck"], sys.modules["spatialmath.base.symbolic"] = smb.argcheck, smb.symbolic
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
        if frne is not None:
            _load_dh(ns, rtb, tools, robot, frne)
    finally:
        for k, v in saved.items():          # the classes keep their own references; the stand-ins must not leak to other importers
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _LOADED[tag] = ns
    return ns


def _load_dh(ns, rtb, tools, robot, frne):
    """robot/Link.py ... robot/DHRobot.py and models/DH/{Puma560,Panda}.py executed unmodified; `roboticstoolbox.frne` is `frne`."""
    sys.modules["roboticstoolbox.frne"] = frne
    rtb.frne = frne
    rtb.rtb_set_param = lambda *a, **k: None
    for name in _PLACEHOLDERS:
        m = _Placeholders(name)
        m.__path__ = []
        sys.modules[name] = m
    sys.modules["roboticstoolbox.tools.params"].rtb_get_param = rtb.rtb_get_param
    sys.modules["spatialgeometry"].SceneNode, sys.modules["spatialgeometry"].SceneGroup = _SceneNode, _SceneGroup
    sys.modules["roboticstoolbox.backends"].load_backend = lambda *a, **k: None
    import pathlib
    sys.modules["roboticstoolbox.tools.data"].rtb_path_to_datafile = lambda *a, **k: pathlib.PurePosixPath("/rtb-data-not-installed", *[str(x) for x in a])
    tools.xacro, tools.URDF = sys.modules["roboticstoolbox.tools.xacro"], sys.modules["roboticstoolbox.tools.URDF"]
    rtb.backends = sys.modules["roboticstoolbox.backends"]
    mods = {}
    for rel in [f for f in DH_FILES if f.startswith("robot/")]:
        short = rel[:-3].split("/")[-1]
        mods[short] = _exec("roboticstoolbox.robot." + short, rel)
        setattr(robot, short, mods[short])
        if short == "Link":
            rtb.Link, rtb.Link2, rtb.BaseLink = mods[short].Link, mods[short].Link2, mods[short].BaseLink
        if short == "DHLink":
            for nm in ("DHLink", "RevoluteDH", "PrismaticDH", "RevoluteMDH", "PrismaticMDH"):
                setattr(rtb, nm, getattr(mods[short], nm))
        if short == "Gripper":
            rtb.Gripper = mods[short].Gripper
        if short == "Robot":
            rtb.Robot, rtb.Robot2 = mods[short].Robot, mods[short].Robot2
    rtb.DHRobot = mods["DHRobot"].DHRobot
    models = types.ModuleType("roboticstoolbox.models")
    models.__path__ = []
    sys.modules["roboticstoolbox.models"] = models
    dh = types.ModuleType("roboticstoolbox.models.DH")
    dh.__path__ = []
    sys.modules["roboticstoolbox.models.DH"] = dh
    models.DH = dh
    rtb.models = models
    dh.Puma560 = _exec("roboticstoolbox.models.DH.Puma560", "models/DH/Puma560.py").Puma560
    dh.Panda = _exec("roboticstoolbox.models.DH.Panda", "models/DH/Panda.py").Panda
    ns.frne, ns.DHRobot, ns.Puma560, ns.PandaDH, ns.mods = frne, rtb.DHRobot, dh.Puma560, dh.Panda, mods
    # the URDF reader, unmodified: tools/urdf/urdf.py lowers a URDF file to the list of Links (URDF.__init__ :1662-1786: one Link per <link> in
    # file order, ets = SE3(origin) RPY [* joint ET], qlim, inertial parameters) that Robot.URDF hands to Robot.__init__ (whose _sort_links numbers
    # the joints).  Its geometry objects (spatialgeometry shapes) are placeholders; meshes are never opened.
    urdf_pkg = types.ModuleType("roboticstoolbox.tools.urdf")
    urdf_pkg.__path__ = []
    sys.modules["roboticstoolbox.tools.urdf"] = urdf_pkg
    tools.urdf = urdf_pkg
    urdf_pkg.utils = _exec("roboticstoolbox.tools.urdf.utils", "tools/urdf/utils.py")
    urdf_pkg.urdf = _exec("roboticstoolbox.tools.urdf.urdf", "tools/urdf/urdf.py")
    ns.URDF = urdf_pkg.urdf.URDF
    ns.Robot = rtb.Robot
    for nm in ("DHLink", "RevoluteDH", "PrismaticDH", "RevoluteMDH", "PrismaticMDH"):
        setattr(ns, nm, getattr(rtb, nm))


def load_dh(fknm, frne, tag):
    """The DH side as well: `ns.DHRobot`, `ns.RevoluteDH` ..., `ns.Puma560()`, `ns.PandaDH()` are the reference's own classes
    (robot/DHRobot.py:35, robot/DHLink.py, robot/Dynamics.py, models/DH/*.py) with `roboticstoolbox.fknm` / `.frne` bound to the given
    modules: `rtbhip.compat.fknm` / `.frne` for the thing under test, the reference's compiled extensions for the checker."""
    return load(fknm, tag, frne=frne)


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

"""Run-time instantiation of the structure-signature kernels (csrc/jit.cpp) -- the Python face of rtbhip_jit_*.

The reference serves every robot through one general code path (core/methods.cpp:318-352, core/ik.cpp:19-75, core/ne.c:62-493,
robot/Robot.py:1704-1903).  Here the straight-line forms of the IK / Newton-Euler / tree-dynamics kernels are instantiated per robot structure:
built into the library for the robots of the benchmarks, compiled by hipRTC on a worker thread for every other robot.  Nothing here is needed
for correct results -- until a robot's code object is ready its calls take the general kernels, which return the same bits; this module is
for callers that want to wait for the fast path (benches, tests) or to look at what happened.
"""
import ctypes as C

from . import _lib

KINDS = {"chain": 0, "dyn": 1, "tree": 2}


def stats():
    """dict of rtbhip_jit_info: available, mode, requested, compiled, disk_hits, failed, pending, launches, general_while_pending,
    compile_seconds(_max), sources, source_digest, last_error."""
    info = _lib.rtbhip_jit_info()
    _lib.check(_lib.lib().rtbhip_jit_stats(C.byref(info)))
    out = {}
    for name, _ in info._fields_:
        v = getattr(info, name)
        out[name] = v.decode(errors="replace") if isinstance(v, bytes) else v
    return out


def wait(timeout=None):
    """Block until no instantiation is queued or being compiled.  True when the queue drained, False when `timeout` seconds ran out."""
    return _lib.lib().rtbhip_jit_wait(-1.0 if timeout is None else float(timeout)) == 0


def _handle_of(obj):
    """(kind, handle) of a mirror object: ETS -> its chain; DHRobot -> its link table; ERobot -> its dynamics tree."""
    if hasattr(obj, "group_table"):                     # ERobot: _handle() is its dynamics tree
        return KINDS["tree"], obj._handle()
    if hasattr(obj, "_dyn_handle"):
        return KINDS["dyn"], obj._dyn_handle()
    if hasattr(obj, "_handle"):
        return KINDS["chain"], obj._handle()
    raise TypeError("jit: expected an ETS, a DHRobot or an ERobot, got %s" % type(obj).__name__)


def prepare(obj, wait_for=False, timeout=None):
    """Ask for ALL run-time instantiations of `obj` now (returns at once unless wait_for).  Initialises the device, unlike constructing the object."""
    kind, h = _handle_of(obj)
    _lib.check(_lib.lib().rtbhip_jit_prepare(kind, h))
    return wait(timeout) if wait_for else True


def names(obj):
    """(name expressions, generated knowledge source) the object's run-time instantiations are requested under; ([], "") when a built-in
    instantiation serves it or no signature applies."""
    kind, h = _handle_of(obj)
    buf = C.create_string_buffer(1 << 16)
    _lib.check(_lib.lib().rtbhip_jit_names(kind, h, buf, len(buf)))
    text = buf.value.decode()
    exprs, _, pre = text.partition("\f")
    return [x for x in exprs.split("\n") if x], pre


def compile_now(unit, expr, arch="gfx950", preamble=""):
    """Compile one instantiation on the calling thread for `arch` (no device needed).  Returns (code_bytes, seconds, from_disk)."""
    cb, sec, fd = C.c_int64(0), C.c_double(0.0), C.c_int32(0)
    u = unit + ("\n" + preamble if preamble else "")
    _lib.check(_lib.lib().rtbhip_jit_compile(u.encode(), expr.encode(), arch.encode(), C.byref(cb), C.byref(sec), C.byref(fd)))
    return cb.value, sec.value, bool(fd.value)

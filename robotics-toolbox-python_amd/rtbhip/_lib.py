"""ctypes binding of librtbhip.so (the C ABI declared in include/rtbhip.h).

This is the stub a reference maintainer would add next to ``from roboticstoolbox.fknm import ...``
(reference robot/ETS.py:28-38): same roles, but a C ABI instead of a CPython extension module.
The library is mandatory: there is NO CPU fallback in this package -- every compute entry point
raises if the HIP library is missing or no GPU is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTBHIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "librtbhip.so")

MEM_HOST, MEM_DEVICE = 0, 1
ET_CONST = 6


class RtbHipError(RuntimeError):
    pass


class rtbhip_et(C.Structure):
    _fields_ = [("kind", C.c_int32), ("flip", C.c_int32), ("jindex", C.c_int32),
                ("reserved", C.c_int32), ("T", C.c_double * 16)]


class rtbhip_tree_group(C.Structure):
    _fields_ = [("parent", C.c_int32), ("kind", C.c_int32), ("flip", C.c_int32), ("jindex", C.c_int32),
                ("T", C.c_double * 16), ("m", C.c_double), ("h", C.c_double * 3), ("I", C.c_double * 6)]


class rtbhip_jit_info(C.Structure):
    _fields_ = [("available", C.c_int32), ("mode", C.c_int32), ("requested", C.c_int64), ("compiled", C.c_int64), ("disk_hits", C.c_int64),
                ("failed", C.c_int64), ("pending", C.c_int64), ("launches", C.c_int64), ("general_while_pending", C.c_int64),
                ("compile_seconds", C.c_double), ("compile_seconds_max", C.c_double), ("sources", C.c_int32), ("source_digest", C.c_char * 20),
                ("last_error", C.c_char * 512)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p
_i64 = C.c_int64
_u64 = C.c_uint64
_i32 = C.c_int32

# name -> (restype, argtypes); must list every symbol include/rtbhip.h declares
SIGNATURES = {
    "rtbhip_last_error": (C.c_char_p, []),
    "rtbhip_version": (C.c_int, []),
    "rtbhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rtbhip_init": (C.c_int, [_i32]),
    "rtbhip_shutdown": (None, []),
    "rtbhip_chain_create": (C.c_int, [C.POINTER(rtbhip_et), _i32, _vp, C.POINTER(_u64)]),
    "rtbhip_chain_create_poe": (C.c_int, [_vp, _i32, _vp, _vp, C.POINTER(_u64)]),
    "rtbhip_chain_upload": (C.c_int, [_u64, _i32]),
    "rtbhip_dyn_upload": (C.c_int, [_u64, _i32]),
    "rtbhip_tree_upload": (C.c_int, [_u64, _i32]),
    "rtbhip_trim": (C.c_int, [_u64, _u64]),
    "rtbhip_chain_destroy": (C.c_int, [_u64]),
    "rtbhip_chain_info": (C.c_int, [_u64, _ip, _ip, _ip]),
    "rtbhip_chain_set_q_width": (C.c_int, [_u64, C.c_int32]),
    "rtbhip_fkine": (C.c_int, [_u64, _vp, _i64, _vp, _vp, _vp, _i32, _vp]),
    "rtbhip_jacob": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_fkine_jacob": (C.c_int, [_u64, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "rtbhip_fkine_jacob_packed": (C.c_int, [_u64, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_hessian": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_hessian_from_jacobian": (C.c_int, [_vp, _i64, _i32, _vp, _i32, _vp]),
    "rtbhip_manipulability_from_jacobian": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _i32, _vp]),
    "rtbhip_jacobm_from_jacobian": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _i32, _vp]),
    "rtbhip_angle_axis": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i32, _vp]),
    "rtbhip_p_servo_error": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _i32, _vp]),
    "rtbhip_p_servo": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, C.c_double, _vp, _vp, _i32, _vp]),
    "rtbhip_ik_lm": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _i32, C.c_double, _i32, _vp, C.c_double,
                               _i32, _i32, _u64, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "rtbhip_ik_lm_nullspace": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _i32, C.c_double, _i32, _vp, C.c_double, _i32, _i32, _u64,
                                         C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "rtbhip_ik_qp": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _i32, C.c_double, _i32, _vp, _u64, C.c_double, C.c_double, C.c_double, C.c_double,
                               C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "rtbhip_ik_target_base": (C.c_int, [_i64]),
    "rtbhip_ik_restart": (C.c_int, [_u64, _u64, _i64, _i32, _vp]),
    "rtbhip_dyn_create": (C.c_int, [_vp, _i32, _i32, C.POINTER(_u64)]),
    "rtbhip_dyn_destroy": (C.c_int, [_u64]),
    "rtbhip_rne": (C.c_int, [_u64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp]),
    "rtbhip_rne_base_wrench": (C.c_int, [_u64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp]),
    "rtbhip_jacob_dot": (C.c_int, [_u64, _vp, _vp, _i64, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_jacob0_dot_analytical": (C.c_int, [_u64, _vp, _vp, _i64, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_jacob0_analytical": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_manipulability": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _i32, _vp, _i32, _vp]),
    "rtbhip_jacobm": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_link_frames": (C.c_int, [_u64, _vp, _i64, _vp, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_partial_fkine0": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _vp, _i32, _vp]),
    "rtbhip_tree_create": (C.c_int, [C.POINTER(rtbhip_tree_group), _i32, C.POINTER(_u64)]),
    "rtbhip_tree_destroy": (C.c_int, [_u64]),
    "rtbhip_tree_rne": (C.c_int, [_u64, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "rtbhip_tree_inertia": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _vp]),
    "rtbhip_tree_coriolis": (C.c_int, [_u64, _vp, _vp, _i64, _vp, _i32, _vp]),
    "rtbhip_tree_accel": (C.c_int, [_u64, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "rtbhip_inertia": (C.c_int, [_u64, _vp, _i64, _vp, _i32, _vp]),
    "rtbhip_coriolis": (C.c_int, [_u64, _vp, _vp, _i64, _vp, _i32, _vp]),
    "rtbhip_accel": (C.c_int, [_u64, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "rtbhip_fleet_fkine_jacob": (C.c_int, [C.POINTER(_u64), _i32, C.POINTER(_vp), C.POINTER(_i64), _i32,
                                           C.POINTER(_vp), C.POINTER(_vp), _i32, _vp]),
    "rtbhip_fleet_fkine_jacob_packed": (C.c_int, [C.POINTER(_u64), _i32, C.POINTER(_vp), C.POINTER(_i64), _i32, C.POINTER(_vp), _i32, _vp]),
    "rtbhip_host_alloc": (C.c_int, [_u64, C.POINTER(_vp)]),
    "rtbhip_host_free": (C.c_int, [_vp]),
    "rtbhip_shard_range": (C.c_int, [_i64, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "rtbhip_device_identity": (C.c_int, [_i32, _vp, _vp]),
    "rtbhip_device_alloc": (C.c_int, [_i32, _u64, C.POINTER(_vp)]),
    "rtbhip_device_free": (C.c_int, [_vp]),
    "rtbhip_device_copy": (C.c_int, [_vp, _vp, _u64, _i32, _vp]),
    "rtbhip_stream_create": (C.c_int, [_i32, C.POINTER(_vp)]),
    "rtbhip_stream_destroy": (C.c_int, [_vp]),
    "rtbhip_stream_sync": (C.c_int, [_vp]),
    "rtbhip_shard_comm_id": (C.c_int, [_vp]),
    "rtbhip_shard_comm_create": (C.c_int, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "rtbhip_shard_comm_create_all": (C.c_int, [_i32, _vp, C.POINTER(_vp)]),
    "rtbhip_shard_comm_destroy": (C.c_int, [_vp]),
    "rtbhip_shard_comm_info": (C.c_int, [_vp, _ip, _ip, _ip]),
    "rtbhip_shard_group": (C.c_int, [_i32]),
    "rtbhip_shard_gather": (C.c_int, [_vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    "rtbhip_last_launch": (C.c_int, [_ip, _ip, _ip]),
    "rtbhip_tune": (C.c_int, [C.c_char_p, _i32]),
    "rtbhip_stream_probe": (C.c_int, [_vp, _i64, _vp, _i64, _vp]),
    "rtbhip_jit_stats": (C.c_int, [C.POINTER(rtbhip_jit_info)]),
    "rtbhip_jit_wait": (C.c_int, [C.c_double]),
    "rtbhip_jit_compile": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_i64), C.POINTER(C.c_double), C.POINTER(_i32)]),
    "rtbhip_jit_prepare": (C.c_int, [_i32, _u64]),
    "rtbhip_jit_names": (C.c_int, [_i32, _u64, C.c_char_p, _i64]),
}

_lib = None


def lib():
    """The loaded library; raises RtbHipError (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtbHipError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); this package has no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64 / libhsa-runtime64, and whichever copy is
        # loaded first serves both (same SONAME).  torch's must be that copy -- it cannot see the GPU through the
        # system runtime -- so bring it in before librtbhip.so pulls in /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().rtbhip_last_error()
        raise RtbHipError("librtbhip error %d: %s" % (rc, msg.decode() if msg else "?"))


def device_count():
    n = C.c_int(0)
    rc = lib().rtbhip_device_count(C.byref(n))
    return n.value if rc == 0 else 0


import contextlib
import threading

_ik_base = threading.local()          # the base this thread has declared (the library keeps its copy per thread too)


@contextlib.contextmanager
def ik_target_base(base):
    """`with rtbhip.ik_target_base(begin): ets.ik_LM(Tep[begin:begin + count], ...)` -- the IK calls inside solve a row block of a
    larger batch and draw the restart vectors the whole batch would draw for those rows (rtbhip_ik_target_base): sharded IK then
    returns, row for row, what one call over all the targets returns.  Nests: leaving the block restores the base that was in
    force when it was entered."""
    prev = getattr(_ik_base, "value", 0)
    check(lib().rtbhip_ik_target_base(int(base)))
    _ik_base.value = int(base)
    try:
        yield
    finally:
        _ik_base.value = prev
        check(lib().rtbhip_ik_target_base(prev))


def tune(key, value):
    check(lib().rtbhip_tune(key.encode(), int(value)))


def last_launch():
    g, b, l = _i32(0), _i32(0), _i32(0)
    lib().rtbhip_last_launch(C.byref(g), C.byref(b), C.byref(l))
    return g.value, b.value, l.value


def device_identity(device=None):
    """{"device", "pci_bus_id", "uuid"} of HIP device `device` (default: torch's current one) -- rtbhip_device_identity."""
    if device is None:
        import torch
        device = torch.cuda.current_device()
    bus, uu = C.create_string_buffer(32), (C.c_ubyte * 16)()
    check(lib().rtbhip_device_identity(int(device), bus, uu))
    return {"device": int(device), "pci_bus_id": bus.value.decode(), "uuid": bytes(uu).hex()}


def shard_range(N, rank, world):
    b, c = _i64(0), _i64(0)
    check(lib().rtbhip_shard_range(int(N), int(rank), int(world), C.byref(b), C.byref(c)))
    return b.value, c.value


# ---------------------------------------------------------------- buffer plumbing
def is_torch(x):
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def host_ptr(a):
    """void* of a C-contiguous float64 / int32 ndarray (or None)."""
    return None if a is None else a.ctypes.data_as(_vp)


def small(a, n):
    """Host-side small parameter (base/tool/gravity/...) -> contiguous float64 array of n elements."""
    if a is None:
        return None
    if is_torch(a):
        a = a.detach().cpu().numpy()
    if hasattr(a, "A") and not isinstance(a, np.ndarray):  # spatialmath SE3-like
        a = a.A
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64)).reshape(-1)
    if a.size != n:
        raise ValueError("expected %d values, got %d" % (n, a.size))
    return a


class _PinnedBlock:
    """One block of rtbhip_host_alloc, exposed through the array interface; goes back to the library's cache when the last
    array viewing it is collected."""

    def __init__(self, nbytes):
        p = _vp()
        check(lib().rtbhip_host_alloc(int(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)
        self.__array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr and _lib is not None:
                _lib.rtbhip_host_free(_vp(self.ptr))
        except Exception:
            pass
        self.ptr = None


PINNED_MIN_BYTES = 1 << 20


def host_empty(shape, dtype=np.float64):
    """Result array of the host path.  From 1 MB up it is PINNED host memory (a cached block of rtbhip_host_alloc): the
    device-to-host DMA then writes straight into it instead of into a staging buffer that a CPU copy has to empty."""
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes < PINNED_MIN_BYTES:
        return np.empty(shape, dtype=dtype)
    try:
        blk = _PinnedBlock(nbytes)
    except RtbHipError:
        return np.empty(shape, dtype=dtype)        # pinning refused (ulimit / fragmentation): a pageable array still works
    return np.asarray(blk).view(dtype).reshape(shape)


def as_numeric(x, what="q"):
    """ndarray(float64) from array-like; TypeError for symbolic/object input -- the reference's
    control-flow signal (core/fknm.cpp:1304-1318 `_check_array_type`, "Symbolic value")."""
    try:
        a = np.asarray(x)
    except Exception:
        raise TypeError("Symbolic value")
    if a.dtype == object or a.dtype.kind not in "fiub":
        raise TypeError("Symbolic value")
    return np.ascontiguousarray(a, dtype=np.float64)


_stream_dev = threading.local()       # the GPU whose current stream the next launch uses (set by note_device)


def note_device(x):
    """Remember which GPU a device-path call's buffers live on: the launch goes to THAT device's current stream (the library switches
    to the buffers' GPU when it is not the current one, rtbhip.h `mem`)."""
    _stream_dev.value = x.device if (is_torch(x) and x.is_cuda) else None


def current_stream_ptr():
    import torch
    dev = getattr(_stream_dev, "value", None)
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def trim(keep_device_bytes=0, keep_pinned_bytes=0):
    """Hand idle cached staging memory back (rtbhip_trim): device blocks of the host-path calls, pinned host blocks."""
    check(lib().rtbhip_trim(int(keep_device_bytes), int(keep_pinned_bytes)))

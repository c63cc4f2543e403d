"""tests/emu_harness.py -- TEST INFRASTRUCTURE.  Python front-end of tests/emu/libemu.so, which runs
the kernels' own __host__ __device__ phase functions lane by lane on the CPU (see tests/emu/emu_common.h).
Used by the `-m "not gpu"` suite to check kernel-body logic against the oracle where no GPU exists."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robotics-toolbox-python_amd")
EMU_SO = os.path.join(ROOT, "tests", "emu", "libemu.so")
SRCS = ["tests/emu/" + f for f in ("emu_kin.cpp", "emu_ik_seq.cpp", "emu_ik_wave_a.cpp", "emu_ik_wave_b.cpp", "emu_ik_wave_c.cpp", "emu_rne.cpp", "emu_dyn_a.cpp", "emu_dyn_b.cpp", "emu_dyn_c.cpp", "emu_dyn_d.cpp", "emu_dyn_e.cpp", "emu_dyn_f.cpp",
                                   "emu_misc.cpp", "emu_tree_big.cpp", "emu_diffjac.cpp")] + ["robotics-toolbox-python_amd/csrc/" + f for f in
                                ("api.cpp", "chain.cpp", "tree.cpp", "hostpipe.cpp", "shard.cpp", "jit.cpp", "kin_kernels.hip", "rne_kernels.hip", "ik_kernels.hip", "dyn_kernels.hip",
                                 "tree_kernels.hip", "tree_dyn_kernels.hip", "partial_kernels.hip", "frames_kernels.hip", "diffjac_kernels.hip")]
_vp, _u64, _i64, _i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_int32
_lib = None


def _deps():
    emu_dir = os.path.join(ROOT, "tests", "emu")
    return sorted(set([os.path.join(ROOT, s) for s in SRCS] + [os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc"))]
                      + [os.path.join(emu_dir, f) for f in os.listdir(emu_dir) if f.endswith((".h", ".cpp"))] + [os.path.join(ROOT, "include", "rtbhip.h")]))


def _digest():
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    return g.source_digest(_deps())


def _stale():
    """By CONTENT (a sha256 of every source kept next to the library), not by file times: those do not survive the trip to the GPU box in a
    useful order, and a rebuild there costs minutes of the GPU lease."""
    if not os.path.exists(EMU_SO):
        return True
    stamp = EMU_SO + ".stamp"
    if os.path.exists(stamp):
        return open(stamp).read().strip() != _digest()
    t = os.path.getmtime(EMU_SO)
    if any(os.path.getmtime(d) > t for d in _deps()):
        return True
    open(stamp, "w").write(_digest())                  # a library from before the stamps existed, newer than its sources: adopt it
    return False


def _deps_newer(obj, dep_file):
    if not os.path.exists(obj) or not os.path.exists(dep_file):
        return True
    t = os.path.getmtime(obj)
    txt = open(dep_file).read().replace("\\\n", " ")
    deps = txt.split(":", 1)[1].split() if ":" in txt else []
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps if not d.startswith("/opt/") and not d.startswith("/usr/"))


def build():
    """tests/emu/*.cpp -- the replay drivers, which only instantiate the kernels' __host__ __device__ bodies -- are compiled HOST side only
    (--cuda-host-only), one object per source in parallel (build/emu is git-ignored; an object is rebuilt when a file it includes
    changed); the library's own sources (api.cpp, chain.cpp, the kernel files with their launchers ...) are not compiled a second
    time: the objects of the product build (build/obj, __graft_entry__.build_lib) are linked in as they are."""
    from concurrent.futures import ThreadPoolExecutor
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build_lib()
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objdir = os.path.join(ROOT, "build", "emu")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-x", "hip", "-w", "-I" + os.path.join(ROOT, "include")]

    def one(src):
        src = os.path.join(ROOT, src)
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        dep = obj + ".d"
        if _deps_newer(obj, dep):
            # the 13..16-joint dynamics bodies are the long poles of this build: -O1 for them (the replay's arithmetic is the same:
            # no fast-math either way, contraction is decided per statement by the front end)
            fl = [f if f != "-O2" else "-O1" for f in flags] if os.path.basename(src) in ("emu_dyn_d.cpp", "emu_dyn_e.cpp", "emu_dyn_f.cpp", "emu_tree_big.cpp") else flags
            subprocess.check_call([hipcc] + fl + ["-MD", "-MF", dep, "-c", src, "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, [x for x in SRCS if x.startswith("tests/emu/")]))
    prod = [os.path.join(ROOT, "build", "obj", os.path.basename(x) + ".o") for x in SRCS if not x.startswith("tests/emu/")]
    prod.append(os.path.join(ROOT, "build", "obj", "jit_sources.cpp.o"))       # generated by the product build (the sources csrc/jit.cpp hands to hipRTC)
    missing = [o for o in prod if not os.path.exists(o)]
    if missing:                                  # a prebuilt library travelled without its objects: compile them
        g.build_lib(force=True)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + prod + ["-o", EMU_SO])
    open(EMU_SO + ".stamp", "w").write(_digest())


def lib():
    global _lib
    if _lib is None:
        if _stale():
            build()
        _lib = C.CDLL(EMU_SO)
        import sys
        sys.path.insert(0, PKG)
        from rtbhip._lib import rtbhip_et
        _lib.rtbhip_chain_create.argtypes = [C.POINTER(rtbhip_et), _i32, _vp, C.POINTER(_u64)]
        _lib.rtbhip_dyn_create.argtypes = [_vp, _i32, _i32, C.POINTER(_u64)]
        _lib.rtbhip_chain_set_q_width.argtypes = [_u64, _i32]
        _lib.emu_kin.argtypes = [_u64, _vp, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _i32]
        _lib.emu_kin_packed.argtypes = [_u64, _vp, _i64, _vp, _vp, _i32, _vp, _i32]
        _lib.emu_pose_mul_seg.argtypes = [_u64, _i32, _vp, _vp, _vp]
        _lib.emu_pose_mul_seg_sig.argtypes = [_u64, _i32, _vp, _vp]
        _lib.emu_chain_signature.argtypes, _lib.emu_chain_signature.restype = [_u64], C.c_uint64
        _lib.emu_rne.argtypes = [_u64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32]
        _lib.emu_rne_base_wrench.argtypes = [_u64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]
        _lib.emu_kin_hess.argtypes = [_u64, _vp, _i64, _vp, _i32, _vp]
        _lib.emu_kin_hess_tile.argtypes = [_u64, _vp, _i64, _vp, _i32, _i32, _vp]
        _lib.emu_diff.argtypes = [_u64, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _vp]
        _lib.emu_ik_nullspace.argtypes = [C.c_double] * 4
        _lib.emu_ik_nullspace.restype = None
        _lib.emu_ik_qp_ks.argtypes = [C.c_double]
        _lib.emu_ik_qp_ks.restype = None
        _lib.emu_ik_target_base.argtypes = [_i64]
        _lib.emu_ik_target_base.restype = None
        _lib.emu_link_frames.argtypes = [_u64, _vp, _i64, _vp, _vp, _i32, _vp]
        _lib.emu_partial.argtypes = [_u64, _vp, _i64, _vp, _i32, _vp]
        _lib.emu_dyn.argtypes = [_u64, _i32, _vp, _vp, _vp, _i64, _vp, _vp]
        _lib.emu_kin_reg.argtypes = [_u64, _vp, _i64, _vp, _vp, _i32, _vp, _vp]
        _lib.emu_sincos.argtypes = [_vp, _i64, _vp, _vp, _i32]
        _lib.emu_ik.argtypes = [_u64, _vp, _i64, _vp, _i32, _i32, C.c_double, _i32, _vp, C.c_double, _i32, _i32, _u64,
                                _vp, _vp, _vp, _vp, _vp]
        _lib.emu_ik_wave.argtypes = [_u64, _i32, _vp, _vp, _i64, _vp, _i32, _i32, C.c_double, _i32, _vp, C.c_double, _i32,
                                     _i32, _u64, _vp, _vp, _vp, _vp, _vp]
        _lib.rtbhip_ik_restart.argtypes = [_u64, _u64, _i64, _i32, _vp]
        _lib.rtbhip_last_error.restype = C.c_char_p
        _lib.emu_hess_from_jac.argtypes = [_vp, _i64, _i32, _vp]
        _lib.emu_angle_axis.argtypes = [_vp, _i64, _vp, _i64, _vp]
        _lib.emu_p_servo_error.argtypes = [_vp, _i64, _vp, _i64, _i32, _vp]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_vp)


def chain_handle(ets):
    """ets: an rtbhip.ETS (its optable() is handed to the emu library's own chain compiler)."""
    from rtbhip._lib import rtbhip_et
    if hasattr(ets, "_twists"):                       # a PoE chain: the emu library's own copy of compile_poe (csrc/chain.cpp)
        ql = np.ascontiguousarray(ets._limits(False).reshape(-1)) if ets.n else None
        tw, T0 = np.ascontiguousarray(ets._twists), np.ascontiguousarray(ets._T0)
        h = _u64(0)
        fn = lib().rtbhip_chain_create_poe
        fn.argtypes = [_vp, _i32, _vp, _vp, C.POINTER(_u64)]
        rc = fn(_p(tw), tw.shape[0], _p(T0), _p(ql), C.byref(h))
        assert rc == 0, lib().rtbhip_last_error()
        return h.value
    rows = ets.optable()
    arr = (rtbhip_et * max(1, len(rows)))()
    for i, (kind, flip, jindex, T) in enumerate(rows):
        arr[i].kind, arr[i].flip, arr[i].jindex = kind, flip, jindex
        flat = np.ascontiguousarray(T, dtype=np.float64).reshape(16)
        for k in range(16):
            arr[i].T[k] = flat[k]
    ql = np.ascontiguousarray(ets._limits(False).reshape(-1)) if ets.n else None
    h = _u64(0)
    rc = lib().rtbhip_chain_create(arr, len(rows), _p(ql), C.byref(h))
    assert rc == 0, lib().rtbhip_last_error()
    if getattr(ets, "_q_width", None) is not None:
        assert lib().rtbhip_chain_set_q_width(h, C.c_int32(ets._q_width)) == 0, lib().rtbhip_last_error()
    return h.value


def sincos(x, reduced_only=False):
    x = np.ascontiguousarray(x, dtype=np.float64)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().emu_sincos(_p(x), x.size, _p(s), _p(c), int(reduced_only))
    return s, c


def kin(ets, q, base=None, tool=None, frame=0, want=("T", "J"), coalesced=True, reg=False):
    """reg=True runs the register-resident variant (kin_reg.h), else the run-time-n tile (kin_tile.h)."""
    h = chain_handle(ets)
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(-1, ets.q_width))
    N, n = q.shape[0], ets.n
    T = np.full((N, 4, 4), np.nan) if "T" in want else None
    J = np.full((N, 6, n), np.nan) if "J" in want else None
    H = np.full((N, n, 6, n), np.nan) if "H" in want else None
    b = None if base is None else np.ascontiguousarray(base, dtype=np.float64)
    t = None if tool is None else np.ascontiguousarray(tool, dtype=np.float64)
    if reg:
        assert H is None
        rc = lib().emu_kin_reg(h, _p(q), N, _p(b), _p(t), frame, _p(T), _p(J))
    else:
        rc = lib().emu_kin(h, _p(q), N, _p(b), _p(t), frame, _p(T), _p(J), _p(H), int(coalesced))
    assert rc == 0
    return T, J, H


def rne(L24, mdh, q, qd, qdd, grav_c, fext=None, force_generic=False):
    L = np.ascontiguousarray(L24, dtype=np.float64).reshape(-1, 24)
    n = L.shape[0]
    h = _u64(0)
    rc = lib().rtbhip_dyn_create(_p(L), n, int(mdh), C.byref(h))
    assert rc == 0, lib().rtbhip_last_error()
    q, qd, qdd = (None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, n)) for x in (q, qd, qdd))
    tau = np.full(q.shape, np.nan)                     # qd / qdd None = NULL pointers: what gravload / itorque hand rtbhip_rne
    g = np.ascontiguousarray(grav_c, dtype=np.float64)
    f = None if fext is None else np.ascontiguousarray(fext, dtype=np.float64)
    rc = lib().emu_rne(h.value, _p(q), _p(qd), _p(qdd), q.shape[0], _p(g), _p(f), _p(tau), int(force_generic))
    assert rc == 0
    return tau


def rne_base_wrench(L24, mdh, q, qd, qdd, grav_c, fext=None):
    """(tau, wbase) as rtbhip_rne_base_wrench's kernel (the run-time-n lane function with the wrench receiver) computes them."""
    L = np.ascontiguousarray(L24, dtype=np.float64).reshape(-1, 24)
    n = L.shape[0]
    h = _u64(0)
    rc = lib().rtbhip_dyn_create(_p(L), n, int(mdh), C.byref(h))
    assert rc == 0, lib().rtbhip_last_error()
    q, qd, qdd = (None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, n)) for x in (q, qd, qdd))
    tau = np.full(q.shape, np.nan)
    wb = np.full((q.shape[0], 6), np.nan)
    g = np.ascontiguousarray(grav_c, dtype=np.float64)
    f = None if fext is None else np.ascontiguousarray(fext, dtype=np.float64)
    rc = lib().emu_rne_base_wrench(h.value, _p(q), _p(qd), _p(qdd), q.shape[0], _p(g), _p(f), _p(tau), _p(wb))
    assert rc == 0
    return tau, wb


def hess_reg(ets, q, tool=None, frame=0, rounds=0):
    """The register-resident Hessian paths on the CPU: rounds=0 k_kin_hess (staged Jacobians + hessian_run),
    rounds=4/8/16 k_kin_hess_tile (per-lane expansion into an LDS tile, whole-wave flush)."""
    h = chain_handle(ets)
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(-1, ets.q_width))
    N, n = q.shape[0], ets.n
    H = np.full((N, n, 6, n), np.nan)
    t = None if tool is None else np.ascontiguousarray(tool, dtype=np.float64)
    if rounds:
        assert lib().emu_kin_hess_tile(h, _p(q), N, _p(t), frame, rounds, _p(H)) == 0
    else:
        assert lib().emu_kin_hess(h, _p(q), N, _p(t), frame, _p(H)) == 0
    return H


def hess_from_jac(J):
    """k_hess_from_jac: (N,6,n) Jacobians -> (N,n,6,n) Hessians."""
    J = np.ascontiguousarray(J, dtype=np.float64)
    N, _, n = J.shape
    H = np.full((N, n, 6, n), np.nan)
    assert lib().emu_hess_from_jac(_p(J), N, n, _p(H)) == 0
    return H


def angle_axis(Te, Tep):
    """k_angle_axis: (N|1,4,4) x (N|1,4,4) -> (N,6)."""
    Te = np.ascontiguousarray(np.asarray(Te, dtype=np.float64).reshape(-1, 4, 4))
    Tep = np.ascontiguousarray(np.asarray(Tep, dtype=np.float64).reshape(-1, 4, 4))
    N = max(len(Te), len(Tep))
    e = np.full((N, 6), np.nan)
    assert lib().emu_angle_axis(_p(Te), len(Te), _p(Tep), len(Tep), _p(e)) == 0
    return e


def p_servo_error(Te, Tep, method):
    """k_angle_axis<method == 1>: (N|1,4,4) x (N|1,4,4) -> (N,6); method 0 angle-axis, 1 rpy (p_servo's default)."""
    Te = np.ascontiguousarray(np.asarray(Te, dtype=np.float64).reshape(-1, 4, 4))
    Tep = np.ascontiguousarray(np.asarray(Tep, dtype=np.float64).reshape(-1, 4, 4))
    N = max(len(Te), len(Tep))
    e = np.full((N, 6), np.nan)
    assert lib().emu_p_servo_error(_p(Te), len(Te), _p(Tep), len(Tep), int(method), _p(e)) == 0
    return e


def diff(ets, mode, q, qd=None, axes=63, tool=None, frame=0):
    """mode 0 jacob_dot (N,6,n), 1 manipulability (N,), 2 jacobm (N,n), 3 jacob0_analytical (N,6,n; axes = representation
    code): diff_device.h on the CPU."""
    h = chain_handle(ets)
    n = ets.n
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(-1, ets.q_width))
    qd = None if qd is None else np.ascontiguousarray(np.asarray(qd, dtype=np.float64).reshape(-1, ets.q_width))
    N = q.shape[0]
    out = np.full({0: (N, 6, n), 1: (N,), 2: (N, n), 3: (N, 6, n), 4: (N, 6, n)}[mode], np.nan)
    t = None if tool is None else np.ascontiguousarray(tool, dtype=np.float64)
    assert lib().emu_diff(h, mode, axes, _p(q), _p(qd), N, _p(t), frame, _p(out)) == 0
    return out


def link_frames(ets, q, marks, base=None):
    """fkine_all through chain.cpp's compile_frames + frames_device.h on the CPU: (N, nmarks, 4, 4)."""
    h = chain_handle(ets)
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(-1, max(ets.q_width, 1)))[:, :ets.q_width]
    q = np.ascontiguousarray(q)
    marks = np.ascontiguousarray(marks, dtype=np.int32)
    out = np.full((q.shape[0], len(marks), 4, 4), np.nan)
    b = None if base is None else np.ascontiguousarray(base, dtype=np.float64)
    assert lib().emu_link_frames(h, _p(q), q.shape[0], _p(b), _p(marks), len(marks), _p(out)) == 0
    return out


def partial(ets, q, order, tool=None):
    """ETS.partial_fkine0 through partial_device.h on the CPU: (N, n, ..., 6, n)."""
    h = chain_handle(ets)
    q = np.ascontiguousarray(np.asarray(q, dtype=np.float64).reshape(-1, ets.q_width))
    N, n = q.shape[0], ets.n
    out = np.full((N,) + (n,) * (order - 1) + (6, n), np.nan)
    t = None if tool is None else np.ascontiguousarray(tool, dtype=np.float64)
    assert lib().emu_partial(h, _p(q), N, _p(t), order, _p(out)) == 0
    return out


def tree_rne(recs, q, qd, qdd, gravity):
    """recs: ERobot.group_table(); runs tree.cpp + tree_device.h on the CPU."""
    from rtbhip._lib import rtbhip_tree_group
    ng = len(recs)
    arr = (rtbhip_tree_group * ng)()
    for k, r in enumerate(recs):
        arr[k].parent, arr[k].kind, arr[k].flip, arr[k].jindex = r["parent"], r["kind"], r["flip"], r["jindex"]
        arr[k].T[:] = list(np.ascontiguousarray(r["T"]).reshape(16))
        arr[k].m = r["m"]
        arr[k].h[:] = list(r["h"])
        arr[k].I[:] = list(r["I"])
    q, qd, qdd = (np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, ng)) for x in (q, qd, qdd))
    tau = np.full(q.shape, np.nan)
    g = np.ascontiguousarray(gravity, dtype=np.float64)
    f = lib().emu_tree_rne
    f.argtypes = [C.POINTER(rtbhip_tree_group), _i32, _vp, _vp, _vp, _i64, _vp, _vp]
    rc = f(arr, ng, _p(q), _p(qd), _p(qdd), q.shape[0], _p(g), _p(tau))
    assert rc == 0, rc
    return tau


def tree_dyn(recs, mode, q, qd=None, torque=None, gravity=None):
    """mode 0 inertia (N,n,n), 1 coriolis (N,n,n), 2 accel (N,n) of an ETS robot: tree_device.h's tree_dyn_lane on the CPU."""
    from rtbhip._lib import rtbhip_tree_group
    ng = len(recs)
    arr = (rtbhip_tree_group * ng)()
    for k, r in enumerate(recs):
        arr[k].parent, arr[k].kind, arr[k].flip, arr[k].jindex = r["parent"], r["kind"], r["flip"], r["jindex"]
        arr[k].T[:] = list(np.ascontiguousarray(r["T"]).reshape(16))
        arr[k].m = r["m"]
        arr[k].h[:] = list(r["h"])
        arr[k].I[:] = list(r["I"])
    q, qd, torque = (None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, ng)) for x in (q, qd, torque))
    N = q.shape[0]
    out = np.full((N, ng) if mode == 2 else (N, ng, ng), np.nan)
    g = None if gravity is None else np.ascontiguousarray(gravity, dtype=np.float64)
    f = lib().emu_tree_dyn
    f.argtypes = [C.POINTER(rtbhip_tree_group), _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]
    rc = f(arr, ng, mode, _p(q), _p(qd), _p(torque), N, _p(g), _p(out))
    assert rc == 0, rc
    return out


def dyn(L24, mdh, mode, q, qd=None, torque=None, grav_c=None):
    """mode 0 inertia (N,n,n), 1 coriolis (N,n,n), 2 accel (N,n): dyn_device.h's per-lane body on the CPU."""
    L = np.ascontiguousarray(L24, dtype=np.float64).reshape(-1, 24)
    n = L.shape[0]
    h = _u64(0)
    assert lib().rtbhip_dyn_create(_p(L), n, int(mdh), C.byref(h)) == 0, lib().rtbhip_last_error()
    conv = lambda x: None if x is None else np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1, n))
    q, qd, torque = conv(q), conv(qd), conv(torque)
    N = q.shape[0]
    out = np.full((N, n) if mode == 2 else (N, n, n), np.nan)
    g = None if grav_c is None else np.ascontiguousarray(grav_c, dtype=np.float64)
    assert lib().emu_dyn(h.value, mode, _p(q), _p(qd), _p(torque), N, _p(g), _p(out)) == 0
    return out


METHODS = {"chan": 0, "wampler": 1, "sugihara": 2, "gn": 3, "nr": 4, "qp": 5}


def ik(ets, Tep, q0=None, ilimit=30, slimit=100, tol=1e-6, joint_limits=True, mask=None, k=1.0, method="chan",
       flavour=0, seed=0, waves=0, stats=None):
    """Runs the IK kernel's per-lane search functions (ik_device.h) on the CPU.  waves == 0: the
    sequential specification (searches of a target one after another); waves > 0: a replay of the
    kernel's per-wave speculative scheduler with that many single-wave workgroups (stats, if a list,
    receives [max wave iterations, total wave iterations, useful lane iterations, scheduling passes])."""
    h = chain_handle(ets)
    n = ets.n
    Tep = np.ascontiguousarray(np.asarray(Tep, dtype=np.float64).reshape(-1, 4, 4))
    N = Tep.shape[0]
    q0a = None if q0 is None else np.ascontiguousarray(np.asarray(q0, dtype=np.float64).reshape(N, n))
    we = None if mask is None else np.ascontiguousarray(mask, dtype=np.float64)
    q = np.full((N, n), np.nan); ok = np.zeros(N, np.int32); it = np.zeros(N, np.int32); se = np.zeros(N, np.int32)
    E = np.zeros(N)
    if waves > 0:
        st = np.zeros(4)
        rc = lib().emu_ik_wave(h, waves, _p(st), _p(Tep), N, _p(q0a), ilimit, slimit, tol, int(joint_limits), _p(we), k,
                               METHODS[method], flavour, seed, _p(q), _p(ok), _p(it), _p(se), _p(E))
        if stats is not None:
            stats[:] = list(st)
    else:
        rc = lib().emu_ik(h, _p(Tep), N, _p(q0a), ilimit, slimit, tol, int(joint_limits), _p(we), k, METHODS[method], flavour,
                          seed, _p(q), _p(ok), _p(it), _p(se), _p(E))
    assert rc == 0, rc
    return q, ok, it, se, E


def ik_nullspace(kq=0.0, km=0.0, ps=0.0, pi=0.3):
    """Null-space terms for the following emu.ik calls (kq <= 0 switches them off again)."""
    pv = np.asarray(pi, dtype=np.float64).reshape(-1)
    lib().emu_ik_nullspace(float(kq), float(km), float(ps), float(pv[0]))
    if pv.size > 1:                                  # one influence distance per joint (robot/IK.py:519-520)
        pv = np.ascontiguousarray(pv)
        fn = lib().emu_ik_nullspace_pi
        fn.argtypes, fn.restype = [_vp, _i32], None
        fn(_p(pv), pv.size)


def ik_qp_ks(ks=1.0):
    """IK_QP's slack gain for the following emu.ik(method="qp", k=kj) calls."""
    lib().emu_ik_qp_ks(float(ks))


def ik_target_base(base=0):
    """Restart-generator key of row 0 for the following emu.ik calls (rtbhip_ik_target_base)."""
    lib().emu_ik_target_base(int(base))


def ik_restart(ets, seed, target, draw):
    h = chain_handle(ets)
    out = np.empty(ets.n)
    assert lib().rtbhip_ik_restart(h, seed, target, draw, _p(out)) == 0
    return out

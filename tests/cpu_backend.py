"""tests/cpu_backend.py -- TEST INFRASTRUCTURE, never importable from the package.

A stand-in for the loaded librtbhip.so that lets the HOST layer of this backend (rtbhip.ET / ETS / DHRobot / ERobot / PoERobot / the compat
modules ...) run where no GPU exists, so that `-m "not gpu"` can exercise the Python mirror of the reference's interface -- shapes, keyword
handling, error behaviour, the reference's own test files -- on the kernels' own arithmetic:

  * every non-compute entry point (chain / dyn / tree handles, the chain compiler, rtbhip_ik_restart, rtbhip_shard_range, rtbhip_tune ...) is
    the REAL one: tests/emu/libemu.so links the product's api.cpp / chain.cpp / tree.cpp objects unchanged;
  * a compute entry point first calls the REAL entry point too: its argument validation runs as on the GPU box and fails with
    RTBHIP_EINVAL / RTBHIP_ELIMIT and the product's message where the product would; where validation passes, the real function stops at its first
    HIP call (no device here: RTBHIP_EHIP) and ONLY THEN the kernel's __host__ __device__ body is replayed lane by lane on the CPU
    (tests/emu/*.cpp, the same replay tests/test_kernel_emu.py checks against the oracle);
  * only host buffers (RTBHIP_MEM_HOST) are served; a device-path call raises -- except under the replay of the GPU suite
    (RTBHIP_TEST_CPU_REPLAY=1, tests/conftest.py), where "device" tensors are CPU tensors wearing the device label and a RTBHIP_MEM_DEVICE
    call is served from the host memory behind them, so that the host layer's device-tensor branch runs too.

The product has no such path: rtbhip._lib.lib() loads librtbhip.so or raises, and nothing under robotics-toolbox-python_amd/ knows this file.  It is
installed by the `cpu_backend` fixture below (tests only), which swaps rtbhip._lib._lib for the duration of a test module."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "robotics-toolbox-python_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import emu_harness                                            # noqa: E402
from rtbhip import _lib as L                                  # noqa: E402

_vp, _u64, _i64, _i32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_int32
EINVAL, EHIP, ELIMIT = -1, -2, -3


def _val(x):
    return x.value if hasattr(x, "value") else x


class EmuBackend:
    """Quacks like the ctypes handle of librtbhip.so."""

    def __init__(self):
        self.emu = emu_harness.lib()
        for name, (res, args) in L.SIGNATURES.items():        # the real entry points, as rtbhip._lib declares them
            fn = getattr(self.emu, name)
            fn.restype, fn.argtypes = res, args
        e = self.emu
        e.emu_tree_rne.argtypes = [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp]
        e.emu_tree_dyn.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]
        e.emu_ik_nullspace_pi.argtypes, e.emu_ik_nullspace_pi.restype = [_vp, _i32], None
        e.emu_diff_from_jac.argtypes = [_i32, _i32, _vp, _vp, _i64, _i32, _vp]
        self._trees = {}                                      # handle -> (group records as bytes, ng)
        self.calls = {}                                       # entry point -> number of replays (the tests assert the replay really ran)

    # ------------------------------------------------------------------ plumbing
    def __getattr__(self, name):                              # everything not overridden below: the real function
        return getattr(self.emu, name)

    def _gate(self, name, args):
        """The real entry point's verdict on the arguments: None = go on and replay, else the code to return."""
        mem = args[-2]
        if _val(mem) != L.MEM_HOST:
            if not getattr(self, "device_label_is_host", False):
                raise L.RtbHipError("tests/cpu_backend.py serves host buffers only: the device path needs a GPU")
            args = tuple(args[:-2]) + (L.MEM_HOST, None)        # tests/conftest.py: CPU tensors wearing the device label (replay of the GPU suite)
        rc = getattr(self.emu, name)(*args)
        if rc == EHIP:
            self.calls[name] = self.calls.get(name, 0) + 1
            return None
        return rc                                             # 0 (nothing to do: N == 0) or the validation error

    def _chain_n(self, h):
        n, m, w = _i32(0), _i32(0), _i32(0)
        self.emu.rtbhip_chain_info(h, C.byref(n), C.byref(m), C.byref(w))
        return n.value, w.value

    # ------------------------------------------------------------------ housekeeping that would need the device
    def rtbhip_init(self, n):
        return 0

    def rtbhip_chain_upload(self, h, dev):
        return 0

    rtbhip_dyn_upload = rtbhip_tree_upload = rtbhip_chain_upload

    def rtbhip_trim(self, a, b):
        return 0

    def rtbhip_tree_create(self, groups, ng, out):
        rc = self.emu.rtbhip_tree_create(groups, ng, out)
        if rc == 0:
            ng = _val(ng)
            self._trees[_val(out._obj)] = (C.string_at(groups, C.sizeof(L.rtbhip_tree_group) * ng), ng)
        return rc

    def rtbhip_ik_target_base(self, base):
        rc = self.emu.rtbhip_ik_target_base(base)
        if rc == 0:
            self.emu.emu_ik_target_base(_val(base))
        return rc

    # ------------------------------------------------------------------ the sharded path's one exchange (csrc/shard.cpp) -- REPLAY ONLY, a world of one:
    # the real entry points' argument checks first; the communicator is a token, the "gather" the copy a single rank's exchange is
    def rtbhip_shard_comm_id(self, buf):
        C.memset(buf, 0x5a, 128)
        return 0

    def rtbhip_shard_comm_create(self, id128, world, rank, out):
        if _val(world) != 1 or _val(rank) != 0:
            raise L.RtbHipError("tests/cpu_backend.py replays a world of one rank only")
        out._obj.value = 0x1234
        return 0

    def rtbhip_shard_comm_destroy(self, comm):
        return 0

    def rtbhip_shard_comm_info(self, comm, w, r, v):
        for ref, val in ((w, 1), (r, 0), (v, 22000)):
            if ref:
                ref._obj.value = val
        return 0

    def rtbhip_shard_gather(self, comm, local, rows, row_bytes, N, world, rank, root, out, stream):
        rc = self.emu.rtbhip_shard_gather(None, local, rows, row_bytes, N, world, rank, root, out, stream)      # the real checks (comm = NULL: allowed for one rank)
        if rc == EHIP:                                                                                        # ... which then stop at the device copy
            C.memmove(out, local, _val(N) * _val(row_bytes))
            return 0
        return rc

    # ------------------------------------------------------------------ kinematics
    def rtbhip_fkine(self, h, q, N, base, tool, T, mem, stream):
        rc = self._gate("rtbhip_fkine", (h, q, N, base, tool, T, mem, stream))
        return self.emu.emu_kin(h, q, N, base, tool, 0, T, None, None, 1) if rc is None else rc

    def rtbhip_jacob(self, h, q, N, tool, frame, J, mem, stream):
        rc = self._gate("rtbhip_jacob", (h, q, N, tool, frame, J, mem, stream))
        return self.emu.emu_kin(h, q, N, None, tool, frame, None, J, None, 1) if rc is None else rc

    def rtbhip_fkine_jacob(self, h, q, N, base, tool, frame, T, J, mem, stream):
        rc = self._gate("rtbhip_fkine_jacob", (h, q, N, base, tool, frame, T, J, mem, stream))
        return self.emu.emu_kin(h, q, N, base, tool, frame, T, J, None, 1) if rc is None else rc

    def rtbhip_fkine_jacob_packed(self, h, q, N, base, tool, frame, TJ, mem, stream):
        rc = self._gate("rtbhip_fkine_jacob_packed", (h, q, N, base, tool, frame, TJ, mem, stream))
        return self.emu.emu_kin_packed(h, q, N, base, tool, frame, TJ, 1) if rc is None else rc

    def rtbhip_hessian(self, h, q, N, tool, frame, H, mem, stream):
        rc = self._gate("rtbhip_hessian", (h, q, N, tool, frame, H, mem, stream))
        return self.emu.emu_kin(h, q, N, None, tool, frame, None, None, H, 1) if rc is None else rc

    def rtbhip_hessian_from_jacobian(self, J, N, n, H, mem, stream):
        rc = self._gate("rtbhip_hessian_from_jacobian", (J, N, n, H, mem, stream))
        return self.emu.emu_hess_from_jac(J, N, n, H) if rc is None else rc

    def rtbhip_manipulability_from_jacobian(self, J, N, n, axes, method, m, mem, stream):
        rc = self._gate("rtbhip_manipulability_from_jacobian", (J, N, n, axes, method, m, mem, stream))
        return self.emu.emu_diff_from_jac(0, n, J, None, N, (_val(axes) & 63) | (_val(method) << 8), m) if rc is None else rc

    def rtbhip_jacobm_from_jacobian(self, J, H, N, n, axes, Jm, mem, stream):
        rc = self._gate("rtbhip_jacobm_from_jacobian", (J, H, N, n, axes, Jm, mem, stream))
        return self.emu.emu_diff_from_jac(2 if _val(H) else 1, n, J, H, N, _val(axes) & 63, Jm) if rc is None else rc

    def rtbhip_angle_axis(self, Te, nTe, Tep, nTep, e, mem, stream):
        rc = self._gate("rtbhip_angle_axis", (Te, nTe, Tep, nTep, e, mem, stream))
        return self.emu.emu_angle_axis(Te, nTe, Tep, nTep, e) if rc is None else rc

    def rtbhip_p_servo_error(self, Te, nTe, Tep, nTep, method, e, mem, stream):
        rc = self._gate("rtbhip_p_servo_error", (Te, nTe, Tep, nTep, method, e, mem, stream))
        return self.emu.emu_p_servo_error(Te, nTe, Tep, nTep, method, e) if rc is None else rc

    def rtbhip_p_servo(self, Te, nTe, Tep, nTep, method, gain6, threshold, v, arrived, mem, stream):
        rc = self._gate("rtbhip_p_servo", (Te, nTe, Tep, nTep, method, gain6, threshold, v, arrived, mem, stream))
        if rc is not None:
            return rc
        # the kernel's tail on the replayed error vectors (k_angle_axis<.., SERVO>): arrived from e, then the gain
        n = max(_val(nTe), _val(nTep))
        r = self.emu.emu_p_servo_error(Te, nTe, Tep, nTep, method, v)
        if r:
            return r
        e = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(n, 6))
        f = np.ctypeslib.as_array(C.cast(arrived, C.POINTER(C.c_uint8)), shape=(n,))
        g = np.ctypeslib.as_array(C.cast(gain6, C.POINTER(C.c_double)), shape=(6,))
        f[:] = np.abs(e).sum(axis=1) < float(_val(threshold))
        e *= g
        return 0

    def rtbhip_jacob_dot(self, h, q, qd, N, tool, frame, Jd, mem, stream):
        rc = self._gate("rtbhip_jacob_dot", (h, q, qd, N, tool, frame, Jd, mem, stream))
        return self.emu.emu_diff(h, 0, 63, q, qd, N, tool, frame, Jd) if rc is None else rc

    def rtbhip_jacob0_analytical(self, h, q, N, tool, rep, Ja, mem, stream):
        rc = self._gate("rtbhip_jacob0_analytical", (h, q, N, tool, rep, Ja, mem, stream))
        return self.emu.emu_diff(h, 3, rep, q, None, N, tool, 0, Ja) if rc is None else rc

    def rtbhip_jacob0_dot_analytical(self, h, q, qd, N, tool, rep, Jd, mem, stream):
        rc = self._gate("rtbhip_jacob0_dot_analytical", (h, q, qd, N, tool, rep, Jd, mem, stream))
        return self.emu.emu_diff(h, 4, rep, q, qd, N, tool, 0, Jd) if rc is None else rc

    def rtbhip_manipulability(self, h, q, N, tool, axes, method, m, mem, stream):
        rc = self._gate("rtbhip_manipulability", (h, q, N, tool, axes, method, m, mem, stream))
        return self.emu.emu_diff(h, 1, (_val(axes) & 63) | (_val(method) << 8), q, None, N, tool, 0, m) if rc is None else rc

    def rtbhip_jacobm(self, h, q, N, tool, axes, Jm, mem, stream):
        rc = self._gate("rtbhip_jacobm", (h, q, N, tool, axes, Jm, mem, stream))
        return self.emu.emu_diff(h, 2, axes, q, None, N, tool, 0, Jm) if rc is None else rc

    def rtbhip_link_frames(self, h, q, N, base, marks, nmarks, out, mem, stream):
        rc = self._gate("rtbhip_link_frames", (h, q, N, base, marks, nmarks, out, mem, stream))
        return self.emu.emu_link_frames(h, q, N, base, marks, nmarks, out) if rc is None else rc

    def rtbhip_partial_fkine0(self, h, q, N, tool, order, out, mem, stream):
        rc = self._gate("rtbhip_partial_fkine0", (h, q, N, tool, order, out, mem, stream))
        return self.emu.emu_partial(h, q, N, tool, order, out) if rc is None else rc

    def rtbhip_fleet_fkine_jacob(self, chains, nc, q, N, frame, T, J, mem, stream):
        rc = self._gate("rtbhip_fleet_fkine_jacob", (chains, nc, q, N, frame, T, J, mem, stream))
        if rc is not None:
            return rc
        for i in range(_val(nc)):                              # k_fleet runs the run-time-n tile per robot; so does this
            Ti = T[i] if T else None
            Ji = J[i] if J else None
            r = self.emu.emu_kin(chains[i], q[i], N[i], None, None, frame, Ti, Ji, None, 1)
            if r:
                return r
        return 0

    def rtbhip_fleet_fkine_jacob_packed(self, chains, nc, q, N, frame, TJ, mem, stream):
        rc = self._gate("rtbhip_fleet_fkine_jacob_packed", (chains, nc, q, N, frame, TJ, mem, stream))
        if rc is not None:
            return rc
        for i in range(_val(nc)):                              # k_fleet<CLS, true>: the register tile to 10 joints, the run-time-n tile beyond
            r = self.emu.emu_kin_packed(chains[i], q[i], N[i], None, None, frame, TJ[i], 1)
            if r:
                return r
        return 0

    # ------------------------------------------------------------------ inverse kinematics
    def _ik(self, name, args, h, Tep, N, q0, ilimit, slimit, tol, rj, we, lam, method, flavour, seed, kq, km, ps, pi, ks, outs):
        rc = self._gate(name, args)
        if rc is not None:
            return rc
        n, _ = self._chain_n(h)
        e = self.emu
        e.emu_ik_nullspace(float(_val(kq)), float(_val(km)), float(_val(ps)), 0.3)      # pi = NULL: the reference's default 0.3
        if pi:
            e.emu_ik_nullspace_pi(pi, n)
        e.emu_ik_qp_ks(float(_val(ks)))
        try:
            return e.emu_ik(h, Tep, N, q0, ilimit, slimit, tol, rj, we, lam, method, flavour, seed, *outs)
        finally:
            e.emu_ik_nullspace(0.0, 0.0, 0.0, 0.3)
            e.emu_ik_qp_ks(1.0)

    def rtbhip_ik_lm(self, h, Tep, N, q0, il, sl, tol, rj, we, lam, method, flavour, seed, qo, ok, it, se, res, mem, stream):
        a = (h, Tep, N, q0, il, sl, tol, rj, we, lam, method, flavour, seed, qo, ok, it, se, res, mem, stream)
        return self._ik("rtbhip_ik_lm", a, h, Tep, N, q0, il, sl, tol, rj, we, lam, method, flavour, seed, 0.0, 0.0, 0.1, None, 1.0,
                        (qo, ok, it, se, res))

    def rtbhip_ik_lm_nullspace(self, h, Tep, N, q0, il, sl, tol, rj, we, lam, method, flavour, seed, kq, km, ps, pi, qo, ok, it, se, res,
                               mem, stream):
        a = (h, Tep, N, q0, il, sl, tol, rj, we, lam, method, flavour, seed, kq, km, ps, pi, qo, ok, it, se, res, mem, stream)
        return self._ik("rtbhip_ik_lm_nullspace", a, h, Tep, N, q0, il, sl, tol, rj, we, lam, method, flavour, seed, kq, km, ps, pi, 1.0,
                        (qo, ok, it, se, res))

    def rtbhip_ik_qp(self, h, Tep, N, q0, il, sl, tol, rj, we, seed, kj, ks, kq, km, ps, pi, qo, ok, it, se, res, mem, stream):
        a = (h, Tep, N, q0, il, sl, tol, rj, we, seed, kj, ks, kq, km, ps, pi, qo, ok, it, se, res, mem, stream)
        return self._ik("rtbhip_ik_qp", a, h, Tep, N, q0, il, sl, tol, rj, we, kj, 5, 1, seed, kq, km, ps, pi, ks, (qo, ok, it, se, res))

    # ------------------------------------------------------------------ dynamics of DH chains
    def rtbhip_rne(self, h, q, qd, qdd, N, g, f, tau, mem, stream):
        rc = self._gate("rtbhip_rne", (h, q, qd, qdd, N, g, f, tau, mem, stream))
        return self.emu.emu_rne(h, q, qd, qdd, N, g, f, tau, 0) if rc is None else rc

    def rtbhip_rne_base_wrench(self, h, q, qd, qdd, N, g, f, tau, wb, mem, stream):
        rc = self._gate("rtbhip_rne_base_wrench", (h, q, qd, qdd, N, g, f, tau, wb, mem, stream))
        return self.emu.emu_rne_base_wrench(h, q, qd, qdd, N, g, f, tau, wb) if rc is None else rc

    def rtbhip_inertia(self, h, q, N, M, mem, stream):
        rc = self._gate("rtbhip_inertia", (h, q, N, M, mem, stream))
        return self.emu.emu_dyn(h, 0, q, None, None, N, None, M) if rc is None else rc

    def rtbhip_coriolis(self, h, q, qd, N, Cm, mem, stream):
        rc = self._gate("rtbhip_coriolis", (h, q, qd, N, Cm, mem, stream))
        return self.emu.emu_dyn(h, 1, q, qd, None, N, None, Cm) if rc is None else rc

    def rtbhip_accel(self, h, q, qd, tq, N, g, qdd, mem, stream):
        rc = self._gate("rtbhip_accel", (h, q, qd, tq, N, g, qdd, mem, stream))
        return self.emu.emu_dyn(h, 2, q, qd, tq, N, g, qdd) if rc is None else rc

    # ------------------------------------------------------------------ dynamics of link trees
    def _tree(self, h):
        rec, ng = self._trees[_val(h)]
        return C.cast(C.c_char_p(rec), _vp), ng

    def rtbhip_tree_rne(self, h, q, qd, qdd, N, g, tau, mem, stream):
        rc = self._gate("rtbhip_tree_rne", (h, q, qd, qdd, N, g, tau, mem, stream))
        if rc is not None:
            return rc
        rec, ng = self._tree(h)
        zeros = np.zeros(max(1, _val(N)) * ng)                # NULL qdd = zeros (gravload): the replay driver wants an array
        qd = qd if _val(qd) else None                         # NULL qd stays NULL: the at-rest instantiation, as the product dispatches
        qdd = qdd if _val(qdd) else zeros.ctypes.data_as(_vp)
        self.emu.emu_tree_rne.argtypes = [_vp, _i32, _vp, _vp, _vp, _i64, _vp, _vp]      # (tests/emu_harness.py re-types it per call)
        return self.emu.emu_tree_rne(rec, ng, q, qd, qdd, N, g, tau)

    def _tree_dyn(self, name, args, h, mode, q, qd, tq, N, g, out):
        rc = self._gate(name, args)
        if rc is not None:
            return rc
        rec, ng = self._tree(h)
        self.emu.emu_tree_dyn.argtypes = [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _vp, _vp]
        return self.emu.emu_tree_dyn(rec, ng, mode, q, qd, tq, N, g, out)

    def rtbhip_tree_inertia(self, h, q, N, M, mem, stream):
        return self._tree_dyn("rtbhip_tree_inertia", (h, q, N, M, mem, stream), h, 0, q, None, None, N, None, M)

    def rtbhip_tree_coriolis(self, h, q, qd, N, Cm, mem, stream):
        return self._tree_dyn("rtbhip_tree_coriolis", (h, q, qd, N, Cm, mem, stream), h, 1, q, qd, None, N, None, Cm)

    def rtbhip_tree_accel(self, h, q, qd, tq, N, g, qdd, mem, stream):
        return self._tree_dyn("rtbhip_tree_accel", (h, q, qd, tq, N, g, qdd, mem, stream), h, 2, q, qd, tq, N, g, qdd)


_backend = None


def backend():
    global _backend
    if _backend is None:
        _backend = EmuBackend()
    return _backend


class installed:
    """`with cpu_backend.installed() as be:` -- rtbhip._lib.lib() hands out the stand-in inside the block, the previous state after it."""

    def __enter__(self):
        self.prev = L._lib
        L._lib = backend()
        return L._lib

    def __exit__(self, *exc):
        L._lib = self.prev
        return False

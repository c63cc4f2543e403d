"""Structure signatures of DH / modified-DH chains (csrc/rne_device.h: RneSig, kRneSigPanda, kRneSigPuma560; rne_kernels.hip's dispatch).

`k_rne` instantiated for a signature reads no shortcut flag and takes no wave-uniform branch on one, rotates about x in the form of alpha's class
(alpha = 0: identity; sin alpha = +-1 exactly: one fused multiply-add per component), drops the cross-product terms of the zero components of
p* = (a, -+d sin alpha, d cos alpha) and, for a robot without friction / motor inertia, the terms that multiply them.  Everything dropped is an
exact zero of the general recursion (cos(pi/2) = 6.1e-17 stays): the specialised kernels agree with the general ones (rtbhip_tune "rne_sig" = 0) to
rounding, and with the oracle exactly as the general ones do (tests/test_kernel_emu.py, test_00_gpu_parity.py run with the switch on).

`-m "not gpu"`: the kernel body on the CPU replay (tests/emu mirrors the launcher's dispatch); `-m gpu`: the kernels."""
import ctypes as C

import numpy as np
import numpy.testing as nt
import pytest

import rtbhip
from oracle import oracle

PANDA, PUMA = 0xe00047a99a2c7ea9, 0x80000377f646a533


def _sig(rob):
    import emu_harness as emu
    f = emu.lib().emu_rne_signature
    f.argtypes, f.restype = [C.c_uint64], C.c_uint64
    h = rob._dyn_handle()
    return f(h.value if hasattr(h, "value") else int(h))


def _models():
    return {"Panda": rtbhip.models.DH.Panda(), "Puma560": rtbhip.models.DH.Puma560()}


def test_shipped_models_have_the_instantiated_signatures():
    import cpu_backend
    with cpu_backend.installed():
        m = _models()
        assert _sig(m["Panda"]) == PANDA and _sig(m["Puma560"]) == PUMA
        # another robot: one link's offset changed -> the same signature (q offsets are not part of it); one alpha changed -> another
        p2 = rtbhip.models.DH.Panda()
        p2.links[2].alpha = 1.0
        p2.dynchanged()
        assert _sig(p2) not in (0, PANDA) and (_sig(p2) >> 63) == 1
        p3 = rtbhip.models.DH.Panda()
        p3.links[3].B = 0.1
        p3.dynchanged()
        assert (_sig(p3) >> 62) & 1 == 0                              # friction: not the instantiated signature
        for j in range(7):
            fields = (PANDA >> (7 * j)) & 127
            assert fields & 1                                           # centres of mass at the link origins


def _calls(rob, q, qd, qdd, g, fext):
    return {"rne": rob.rne(q, qd, qdd, gravity=g), "rne_fext": rob.rne(q, qd, qdd, gravity=g, fext=fext), "gravload": rob.gravload(q, gravity=g),
            "itorque": rob.itorque(q, qdd), "rne_noqdd": rob.rne(q, qd, None, gravity=g),
            "inertia": rob.inertia(q), "coriolis": rob.coriolis(q, qd), "accel": rob.accel(q, qd, qdd, gravity=g)}


def _both(rob, N, seed):
    rng = np.random.default_rng(seed)
    n = rob.n
    q, qd, qdd = rng.uniform(-3, 3, (N, n)), rng.normal(size=(N, n)), rng.normal(size=(N, n))
    g, fext = np.array([0.3, -0.2, 9.81]), rng.normal(size=6)
    out = {}
    try:
        for s in (1, 0):
            rtbhip.tune("rne_sig", s)
            out[s] = {k: np.asarray(v) for k, v in _calls(rob, q, qd, qdd, g, fext).items()}
    finally:
        rtbhip.tune("rne_sig", 1)
    return out, (q, qd, qdd, g, fext)


def _compare(out, exact=False):
    for k in out[1]:
        a, b = out[1][k], out[0][k]
        assert np.isfinite(b).all()
        if exact:
            nt.assert_array_equal(a, b, err_msg=k)
        else:
            nt.assert_allclose(a, b, rtol=0, atol=(1e-9 if k == "accel" else 1e-12) * max(1.0, np.abs(b).max()), err_msg=k)


@pytest.mark.parametrize("name", ["Panda", "Puma560"])
def test_signature_kernel_body_equals_the_general_one_and_the_oracle(name):
    import cpu_backend
    with cpu_backend.installed():
        rob = _models()[name]
        out, (q, qd, qdd, g, fext) = _both(rob, 60, 5)
    _compare(out)
    mdh = 1 if name == "Panda" else 0
    for key, args in (("rne", (q, qd, qdd, -g)), ("rne_fext", (q, qd, qdd, -g, fext)), ("gravload", (q, 0 * q, 0 * q, -g)), ("itorque", (q, 0 * q, qdd, np.zeros(3)))):
        want = oracle.rne_dh(rob.L24(), mdh, *args)
        nt.assert_allclose(out[1][key], want, rtol=0, atol=1e-11 * max(1.0, np.abs(want).max()), err_msg=key)
    # the mixin terms against the oracle's restated loops (oracle.inertia_dh ..., pinned on the reference's own mixin: test_dynamics_terms.py)
    k = slice(0, 5)
    L = rob.L24()
    for key, want in (("inertia", oracle.inertia_dh(L, mdh, q[k])), ("coriolis", oracle.coriolis_dh(L, mdh, q[k], qd[k])),
                      ("accel", oracle.accel_dh(L, mdh, q[k], qd[k], qdd[k], -g))):
        nt.assert_allclose(out[1][key][k], want, rtol=0, atol=(1e-8 if key == "accel" else 1e-11) * max(1.0, np.abs(want).max()), err_msg=key)


def test_switch_leaves_a_robot_with_another_signature_alone():
    import cpu_backend
    with cpu_backend.installed():
        rob = rtbhip.models.DH.Panda()
        rob.links[2].alpha = 1.0
        rob.dynchanged()
        out, _ = _both(rob, 20, 6)
    _compare(out, exact=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Panda", "Puma560"])
def test_gpu_signature_kernels_equal_the_general_kernels(name):
    out, _ = _both(_models()[name], 20000, 7)
    _compare(out)

"""rtb.p_servo (reference tools/p_servo.py:46-117) on this backend, both methods -- "rpy" is the reference's DEFAULT.

The checker is the reference's OWN p_servo function, loaded unmodified (oracle/ref_classes.py) over the stand-in spatialmath whose
`base.tr2rpy` is the oracle's restatement of the third-party routine (oracle/poe.py: tr2rpy_zyx).  That restatement is pinned here by (i) the
literal of the reference's tests/test_tools.py:36-51, (ii) reconstruction R == Rz(yaw) Ry(pitch) Rx(roll) on random and on singular
(|pitch| = pi/2) rotations, and elsewhere by the PoE tests (the reference's twist -> ETS recipe goes through SE3.rpy()).
`-m "not gpu"`: the kernel body (k_angle_axis<true> = servo_rpy_lane, csrc/servo_device.h) replayed on the CPU; `-m gpu`: librtbhip.so."""
import math

import numpy as np
import numpy.testing as nt
import pytest

from oracle import poe, ref_classes, ref_harness

needs_ref = pytest.mark.skipif(not (ref_classes.available() and ref_harness.available()),
                               reason="needs oracle/_ref (the reference's byte-compiled tools/p_servo.py)")


def _rot(axis, a):
    c, s = math.cos(a), math.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])}[axis]


def _pose(rpy, t):
    T = np.eye(4)
    T[:3, :3] = _rot("z", rpy[2]) @ _rot("y", rpy[1]) @ _rot("x", rpy[0])
    T[:3, 3] = t
    return T


def pairs(seed=0, n=150):
    """(Te, Tep, tag): random pairs; identical pairs; relative rotations at the tr2rpy singularity (pitch = +-pi/2, exact and one ulp-scale
    step away); relative rotations about one axis only; translation only."""
    rng = np.random.default_rng(seed)
    Te = np.array([_pose(rng.uniform(-np.pi, np.pi, 3) * [1, 0.5, 1], rng.uniform(-1, 1, 3)) for _ in range(n)])
    Tep = np.array([_pose(rng.uniform(-np.pi, np.pi, 3) * [1, 0.5, 1], rng.uniform(-1, 1, 3)) for _ in range(n)])
    tag = np.zeros(n, dtype=int)
    Tep[100:110] = Te[100:110]; tag[100:110] = 1
    for i in range(110, 130):                                  # eTep exactly Ry(+-pi/2) Rx(r) / nearly so, from the identity pose
        Te[i] = np.eye(4)
        p = (np.pi / 2 if i % 2 else -np.pi / 2) + (0.0 if i < 120 else rng.uniform(-3e-8, 3e-8))
        Tep[i] = _pose([rng.uniform(-1, 1), p, rng.uniform(-1, 1)], rng.uniform(-1, 1, 3))
        if i < 120:                                            # make |R20| exactly 1 (the singular branch), the other entries consistent
            R = Tep[i][:3, :3]
            R[2, 0] = round(R[2, 0]); R[0, 0] = R[1, 0] = R[2, 1] = R[2, 2] = 0.0
        tag[i] = 2 if i < 120 else 3
    for i in range(130, 140):
        Tep[i] = Te[i] @ _pose([0, 0, 0], rng.uniform(-1, 1, 3)); tag[i] = 4
    for i in range(140, n):
        Tep[i] = Te[i] @ _pose(np.eye(3)[i % 3] * rng.uniform(-3, 3), [0, 0, 0]); tag[i] = 5
    return Te, Tep, tag


def test_tr2rpy_restatement_reconstructs_the_rotation():
    rng = np.random.default_rng(1)
    for _ in range(200):
        R = _pose(rng.uniform(-np.pi, np.pi, 3), [0, 0, 0])[:3, :3]
        r, p, y = poe.tr2rpy_zyx(R)
        nt.assert_allclose(_rot("z", y) @ _rot("y", p) @ _rot("x", r), R, atol=1e-12)
    for s in (1.0, -1.0):                                      # the singular branch: roll := 0, yaw carries the whole in-plane rotation
        for a in (-2.0, 0.3, 2.9):
            R = _rot("y", s * np.pi / 2) @ _rot("x", a)
            R[2, 0] = -s; R[0, 0] = R[1, 0] = R[2, 1] = R[2, 2] = 0.0
            r, p, y = poe.tr2rpy_zyx(R)
            assert r == 0.0 and abs(abs(p) - np.pi / 2) < 1e-15
            nt.assert_allclose(_rot("z", y) @ _rot("y", p) @ _rot("x", r), R, atol=1e-12)


def _reference_p_servo(ns, Te, Tep, method, **kw):
    out = [ns.p_servo.p_servo(a, b, method=method, **kw) for a, b in zip(Te, Tep)]
    return np.array([o[0] for o in out]), np.array([o[1] for o in out])


@needs_ref
def test_reference_p_servo_under_the_stand_in_reproduces_its_own_test_literal():
    """tests/test_tools.py:36-68 of the reference, both methods, on the reference's own function."""
    ns = ref_classes.load_reference()
    a, b, c = np.eye(4), _pose([0.7, 0, 0], [0, 0, 0]) @ _pose([0, 0, 0], [1, 0, 0]), _pose([0, 0, 0], [0, 0, 0.59])
    for method in ("rpy", "angle-axis"):
        v, arrived = ns.p_servo.p_servo(a, b, method=method)
        nt.assert_array_almost_equal(v, [1, 0, 0, 0.7, 0, 0], decimal=4)
        assert arrived is False and ns.p_servo.p_servo(a, c, threshold=0.6, method=method)[1] is True


def _diff(e, ref):
    """|e - ref| with the three angles compared on the circle: at a relative rotation of exactly pi about one axis atan2(+-0, -x) is
    +pi or -pi by the sign of a zero, and the reference's LU inverse and the kernel's transpose do not produce the same signed zeros."""
    d = e - ref
    d[..., 3:] = (d[..., 3:] + np.pi) % (2 * np.pi) - np.pi
    return np.abs(d)


def _tolerance(tag):
    # near (not at) the singularity the pitch comes out of atan(x / tiny): conditioning, not arithmetic, sets the agreement
    return np.where(tag == 3, 1e-6, 1e-11)


@needs_ref
def test_emu_rpy_kernel_body_equals_the_reference_p_servo():
    import emu_harness as emu
    ns = ref_classes.load_reference()
    Te, Tep, tag = pairs()
    ref, _ = _reference_p_servo(ns, Te, Tep, "rpy")
    e = emu.p_servo_error(Te, Tep, 1)
    assert np.all(_diff(e, ref).max(axis=1) <= _tolerance(tag)), _diff(e, ref).max(axis=1)
    assert np.abs(e[tag == 1][:, :3]).max() == 0.0 and np.abs(e[tag == 1]).max() < 1e-15   # identical poses: R^T R = I to rounding
    assert np.all(e[tag == 2][:, 3] == 0.0)                                        # singular branch: roll is set to zero
    nt.assert_array_equal(emu.p_servo_error(Te, Tep, 0), emu.angle_axis(Te, Tep))  # method 0 is the angle-axis kernel
    for n in (1, 63, 64, 65):
        nt.assert_array_equal(emu.p_servo_error(Te[:n], Tep[:n], 1), e[:n])
    nt.assert_array_equal(emu.p_servo_error(Te[7], Tep[:70], 1), np.array([emu.p_servo_error(Te[7], Tep[i], 1)[0] for i in range(70)]))


@needs_ref
def test_p_servo_host_layer_on_the_cpu_replay():
    """rtbhip.p_servo itself (defaults, gain forms, `arrived`, one pair / stacks, the method strings) over tests/cpu_backend.py."""
    import cpu_backend
    ns = ref_classes.load_reference()
    with cpu_backend.installed():
        _check_p_servo(ns)


def _check_p_servo(ns):
    import rtbhip
    Te, Tep, tag = pairs()
    for method, kw in (("rpy", {}), ("angle-axis", {"method": "angle-axis"})):
        ref_e, ref_a = _reference_p_servo(ns, Te, Tep, method, threshold=2.0)
        e, arrived = rtbhip.p_servo(Te, Tep, threshold=2.0, **kw)                                       # "rpy" is the default, as in the reference
        assert e.shape == (len(Te), 6) and arrived.shape == (len(Te),) and arrived.dtype == bool
        assert np.all(_diff(e, ref_e).max(axis=1) <= _tolerance(tag))
        margin = np.abs(np.abs(ref_e).sum(axis=1) - 2.0) > 1e-5
        nt.assert_array_equal(arrived[margin], ref_a[margin])
        v, _ = rtbhip.p_servo(Te, Tep, gain=[1, 2, 3, 4, 5, 6], **kw)
        nt.assert_array_equal(v, e * [1, 2, 3, 4, 5, 6])
    v1, a1 = rtbhip.p_servo(Te[0], Tep[0], gain=2.0)
    assert v1.shape == (6,) and isinstance(a1, bool)
    nt.assert_allclose(v1, ns.p_servo.p_servo(Te[0], Tep[0], gain=2.0)[0], atol=1e-11)
    a, b, c = np.eye(4), _pose([0.7, 0, 0], [0, 0, 0]) @ _pose([0, 0, 0], [1, 0, 0]), _pose([0, 0, 0], [0, 0, 0.59])
    for kw in ({}, {"method": "rpy"}, {"method": "angle-axis"}):                                       # tests/test_tools.py:36-68
        v, arrived = rtbhip.p_servo(a, b, **kw)
        nt.assert_array_almost_equal(v, [1, 0, 0, 0.7, 0, 0], decimal=4)
        assert arrived is False and rtbhip.p_servo(a, c, threshold=0.6, **kw)[1] is True
    nt.assert_array_equal(rtbhip.angle_axis_python(Te[:5], Tep[:5]), rtbhip.angle_axis(Te[:5], Tep[:5]))
    with pytest.raises(ValueError):
        rtbhip.p_servo(a, b, gain=[1, 2, 3])
    with pytest.raises(rtbhip.RtbHipError):
        rtbhip._lib.check(rtbhip.lib().rtbhip_p_servo_error(None, 1, None, 1, 7, None, 0, None))


@needs_ref
@pytest.mark.gpu
def test_gpu_p_servo_equals_the_reference_p_servo():
    import torch
    import rtbhip
    ns = ref_classes.load_reference()
    _check_p_servo(ns)
    Te, Tep, tag = pairs()
    v, arrived = rtbhip.p_servo(Te, Tep)
    vd, ad = rtbhip.p_servo(torch.from_numpy(Te).cuda(), torch.from_numpy(Tep).cuda())
    nt.assert_array_equal(vd.cpu().numpy(), v)
    nt.assert_array_equal(ad.cpu().numpy(), arrived)
    n = len(Te)
    vb, _ = rtbhip.p_servo(np.tile(Te, (700, 1, 1))[:-3], np.tile(Tep, (700, 1, 1))[:-3])       # 104 997 pairs: many tiles, a ragged last one
    for k in (0, 351, 698):
        nt.assert_array_equal(vb[k * n:(k + 1) * n], v)
    nt.assert_array_equal(vb[699 * n:], v[:n - 3])

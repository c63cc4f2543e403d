"""Structure classes of the constant segments (csrc/chain.cpp: seg_class_bits, csrc/kin_device.h: pose_mul_seg_cls) -- what k_ik multiplies by
instead of a general 3x4 where the folded constant is a pure translation, a quarter turn / a rotation about one axis, or one of the cyclic
permutations of the axis conjugation.  The class is decided from EXACT zeros and ones (cos(pi/2) = 6.1e-17 is kept, nothing is snapped), so the
structured product must equal the general one to the last bit or the one next to it.  Here: every class is produced by a chain a user could
write, recognised, and multiplied right; the Panda of models/ETS/Panda.py:32-54 gets six quarter turns and an axis rotation."""
import ctypes as C

import numpy as np

import emu_harness
import rtbhip
from oracle import chains

NAMES = ["general", "identity", "RxP", "RxN", "Rx", "RyP", "RyN", "Ry", "RzP", "RzN", "Rz", "permA", "permB"]


def classes_of(ets):
    import cpu_backend
    with cpu_backend.installed():                 # chain handles live in the library that made them: make it the replay library
        return _classes_of(ets)


def _classes_of(ets):
    lib = emu_harness.lib()
    h = ets._handle()
    rng = np.random.default_rng(0)
    out = []
    for j in range(ets.n + 1):
        Pin = np.concatenate([np.linalg.qr(rng.normal(size=(3, 3)))[0].ravel(), rng.normal(size=3)])
        a, b = np.zeros(12), np.zeros(12)
        bits = lib.emu_pose_mul_seg(C.c_uint64(h), j, emu_harness._p(Pin), emu_harness._p(a), emu_harness._p(b))
        assert bits >= 0
        # the structured product against the general one: at most a unit in the last place apart (one rounding instead of two in the quarter turns)
        # the structured products against the general one: THE SAME BITS (each form is the general product with its exact zeros dropped and its
        # exact ones taken as the operand, operation for operation -- kin_device.h)
        c = np.zeros(12)
        assert lib.emu_pose_mul_seg_sig(C.c_uint64(h), j, emu_harness._p(Pin), emu_harness._p(c)) == 0
        assert np.array_equal(a, b) and np.array_equal(c, b), (j, NAMES[bits & 15], np.abs(a - b).max(), np.abs(c - b).max())
        out.append((NAMES[bits & 15], bits >> 4))
    return out


def T(name, v):
    return rtbhip.ET.SE3(chains.elementary(name, v))


def test_every_class_is_recognised_and_multiplied_right():
    E = rtbhip.ET
    h = np.pi / 2
    cases = [
        ([E.tz(0.3), E.Rz()], ["identity"]),                                                      # pure translation
        ([E.Rx(h), E.Rz()], ["RxP"]), ([E.Rx(-h), E.tz(0.2), E.Rz()], ["RxN"]), ([E.Rx(0.3), E.tx(0.1), E.Rz()], ["Rx"]),
        ([E.Ry(-h), E.Rz()], ["RyP"]), ([E.Ry(h), E.Rz()], ["RyN"]), ([E.Ry(0.4), E.Rz()], ["Ry"]),
        ([E.Rz(h), E.ty(0.1), E.Rz()], ["RzP"]), ([E.Rz(-h), E.Rz()], ["RzN"]), ([E.Rz(-np.pi / 4), E.Rz()], ["Rz"]),
        ([E.Rx(0.3), E.Ry(0.2), E.Rz()], ["general"]),
        ([E.tx(0.1), E.Rx()], ["permA"]), ([E.ty(0.1), E.Ry()], ["permB"]),                      # the conjugation of an x / y joint
    ]
    seen = set()
    for ets_list, want in cases:
        got = classes_of(rtbhip.ETS(ets_list))
        assert got[0][0] == want[0], (want, got)
        seen |= {g[0] for g in got}
    assert seen >= set(NAMES), set(NAMES) - seen
    # translation masks: exactly the non-zero components
    got = classes_of(rtbhip.ETS([E.tx(0.1), E.tz(-0.2), E.Rz()]))
    assert got[0] == ("identity", 0b101) and got[1] == ("identity", 0)


def test_the_panda_is_six_quarter_turns_and_a_flange_rotation():
    got = classes_of(rtbhip.models.Panda().ets())
    assert [g[0] for g in got] == ["identity", "RxN", "RxP", "RxP", "RxN", "RxP", "RxP", "Rz"], got
    # C_2 = Rx(pi/2) tz(0.316): t = (0, -0.316, 1.9e-17) -- the tiny component is NOT dropped
    assert got[2][1] == 0b110 and got[1][1] == 0 and got[7][1] == 0b100
    # ... which is the signature k_ik has a straight-line instantiation for (ik_kernels.hip: kIkSigPandaETS)
    import cpu_backend
    with cpu_backend.installed():
        ets = rtbhip.models.Panda().ets()                 # (kept alive: the handle dies with its ETS)
        sig = emu_harness.lib().emu_chain_signature(C.c_uint64(ets._handle()))
    cls = {n: i for i, n in enumerate(NAMES)}
    want = 1 << 63
    for j, (name, tm) in enumerate([("identity", 4), ("RxN", 0), ("RxP", 6), ("RxP", 1), ("RxN", 7), ("RxP", 0), ("RxP", 7), ("Rz", 4)]):
        want |= (cls[name] | (tm << 4)) << (7 * j)
    assert sig == want, (hex(sig), hex(want))


def test_random_robots_fkine_through_the_ik_walk_equals_the_oracle():
    """chains of random constants of every kind through ik's FK (first iteration's E at the solution is ~0: ik from the exact q converges at once)"""
    import cpu_backend
    from oracle import oracle
    rng = np.random.default_rng(5)
    with cpu_backend.installed():
        for trial in range(12):
            n = int(rng.integers(3, 8))
            spec = []
            for j in range(n):
                kind = int(rng.integers(0, 5))
                if kind == 0: spec.append((["tx", "ty", "tz"][int(rng.integers(3))], float(rng.uniform(0.05, 0.4))))
                elif kind == 1: spec.append((["Rx", "Ry", "Rz"][int(rng.integers(3))], float(rng.choice([-1, 1])) * np.pi / 2))
                elif kind == 2: spec.append((["Rx", "Ry", "Rz"][int(rng.integers(3))], float(rng.uniform(-1, 1))))
                elif kind == 3: spec += [("Rx", float(rng.uniform(-1, 1))), ("Ry", float(rng.uniform(-1, 1))), ("tx", 0.1)]
                spec.append((["Rz", "Rz", "Rx", "Ry"][int(rng.integers(4))], None, False))
            spec.append(("tz", 0.1))
            from helpers import product_ets
            qlim = np.array([[-2.5] * n, [2.5] * n])
            ets, ch = product_ets(spec, qlim=qlim), chains.Chain(spec, qlim=qlim)
            qs = rng.uniform(-2, 2, (40, n))
            Tep = oracle.fkine(ch, qs)
            q0 = np.clip(qs + 0.05 * rng.normal(size=qs.shape), qlim[0], qlim[1])
            q, ok, it, se, E = ets.ik_LM(Tep, q0=q0, slimit=1, ilimit=50)
            checked = 0
            for i in range(40):
                o = oracle.ik_lm(ch, Tep[i], q0=q0[i], restarts=np.zeros((2, n)), slimit=1, ilimit=50)
                if o[1] and o[3] == 1:
                    checked += 1
                    assert (o[1], o[2], o[3]) == (ok[i], it[i], se[i]), (trial, i)
                    np.testing.assert_allclose(q[i], o[0], atol=1e-6)
            assert checked >= 10, (trial, checked)


def test_signatures_k_ik_is_instantiated_for_are_those_robots_signatures():
    """csrc/ik_kernels.hip: kIkSigPandaETS (also the DH Panda lowered to an ETS), kIkSigPandaURDF, kIkSigUR -- read from the source, compared with what
    the chain compiler computes for the robots they are named after (a constant that drifts from its robot would silently fall back to the general kernel)"""
    import os, re
    import cpu_backend
    from rtbhip import urdf
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robotics-toolbox-python_amd", "csrc", "ik_kernels.hip")).read()
    cls = {"kSeg" + n[0].upper() + n[1:]: i for i, n in enumerate(NAMES)}
    cls.update({"kSegPermA": 11, "kSegPermB": 12})

    def constant(name):
        body = re.search(r"constexpr SegSig %s = (.*?);" % name, src, re.S).group(1)
        sig = 1 << 63
        for j, c, tm in re.findall(r"seg_sig_of\((\d+), (kSeg\w+), (\d+)\)", body):
            sig |= (cls[c] | (int(tm) << 4)) << (7 * int(j))
        return sig
    with cpu_backend.installed():
        lib = emu_harness.lib()
        robots = {"kIkSigPandaETS": [rtbhip.models.Panda().ets(), rtbhip.models.DH.Panda().ets()],
                  "kIkSigPandaURDF": [urdf.load("Panda").ets()],
                  "kIkSigUR": [urdf.load(n).ets(end="tool0") for n in ("UR3", "UR5", "UR10")]}
        for name, chains_ in robots.items():
            for e in chains_:
                assert lib.emu_chain_signature(C.c_uint64(e._handle())) == constant(name), name

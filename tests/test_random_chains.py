"""Randomly generated chains (every transform kind, flips, SE3 constants, leading / trailing / repeated constants, 1..10
joints) through every kinematics output: the chain compiler + kernels against the oracle.  CPU: the kernel bodies replayed
by tests/emu; GPU: through the C ABI."""
import numpy as np
import numpy.testing as nt
import pytest

from oracle import oracle, chains
from helpers import product_ets

AXES = ["Rx", "Ry", "Rz", "tx", "ty", "tz"]


def random_spec(rng, n_joints):
    spec, j = [], 0
    while j < n_joints or rng.random() < 0.5:
        r = rng.random()
        if j < n_joints and r < 0.45:
            spec.append((AXES[rng.integers(6)], None, bool(rng.random() < 0.3)))
            j += 1
        elif r < 0.85:
            a = AXES[rng.integers(6)]
            spec.append((a, float(rng.uniform(-1.5, 1.5) if a[0] == "R" else rng.uniform(-0.4, 0.4))))
        else:
            T = chains.elementary("Rz", rng.uniform(-3, 3)) @ chains.elementary("tx", rng.uniform(-0.3, 0.3)) \
                @ chains.elementary("Ry", rng.uniform(-3, 3)) @ chains.elementary("tz", rng.uniform(-0.3, 0.3))
            spec.append(T)
        if len(spec) > 40:
            break
    while j < n_joints:                                   # the length cap cut it short: finish with joints
        spec.append((AXES[rng.integers(6)], None, False)); j += 1
    return spec


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        n = 1 + k % 10
        spec = random_spec(rng, n)
        out.append((spec, chains.Chain(spec, name="rand%d" % k), rng.uniform(-2.5, 2.5, (5, n)),
                    chains.elementary("tx", rng.uniform(-0.2, 0.2)) @ chains.elementary("Rx", rng.uniform(-1, 1))))
    return out


def test_emu_random_chains():
    import emu_harness as emu
    for spec, ch, q, tool in _cases(7, 40):
        ets = product_ets(spec)
        for reg in (True, False):
            for frame in (0, 1):
                T, J, _ = emu.kin(ets, q, tool=tool, frame=frame, reg=reg)
                nt.assert_allclose(T, oracle.fkine(ch, q, tool=tool), atol=1e-12)
                nt.assert_allclose(J, oracle.jacob(ch, q, tool, frame), atol=1e-12)
        _, _, H = emu.kin(ets, q, tool=tool, want=("H",))
        nt.assert_allclose(H, oracle.hessian(ch, q, tool), atol=1e-12)
        nt.assert_allclose(emu.hess_reg(ets, q, tool=tool, rounds=4), H, atol=1e-13)
        marks = sorted(set(np.random.default_rng(len(spec)).integers(0, ch.m + 1, 4).tolist()))
        nt.assert_allclose(emu.link_frames(ets, q, marks), oracle.link_frames(ch, q, marks), atol=1e-12)


@pytest.mark.gpu
def test_gpu_random_chains():
    rng = np.random.default_rng(1)
    for spec, ch, q, tool in _cases(11, 60):
        ets = product_ets(spec)
        q = np.vstack([q, rng.uniform(-2.5, 2.5, (70, ch.n))])            # more than one tile
        nt.assert_allclose(ets.eval(q, tool=tool), oracle.fkine(ch, q, tool=tool), atol=1e-10)
        nt.assert_allclose(ets.jacob0(q, tool=tool), oracle.jacob(ch, q, tool, 0), atol=1e-10)
        nt.assert_allclose(ets.jacobe(q, tool=tool), oracle.jacob(ch, q, tool, 1), atol=1e-10)
        T, J = ets.fkine_jacob0(q)
        nt.assert_allclose(J, oracle.jacob(ch, q, None, 0), atol=1e-10)
        nt.assert_allclose(ets.hessian0(q[:10], tool=tool), oracle.hessian(ch, q[:10], tool), atol=1e-10)
        qd = rng.normal(size=q.shape)
        nt.assert_allclose(ets.jacob0_dot(q[:10], qd[:10]), oracle.jacob_dot(ch, q[:10], qd[:10]), atol=1e-10)
        marks = list(range(0, ch.m + 1, max(1, ch.m // 6)))[:33]
        nt.assert_allclose(ets.link_frames(q[:10], marks), oracle.link_frames(ch, q[:10], marks), atol=1e-10)

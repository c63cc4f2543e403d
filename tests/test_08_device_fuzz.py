"""The device fuzzers of scripts/ as tests: random robots through EVERY size of every kernel family against the oracle -- the sizes the hand-written
parity tests do not visit one by one (a 2-joint IK, a 13-joint tree, a hand-numbered robot, a modified-DH chain with a prismatic first joint: each
of these found a defect in round 4).  `-m gpu`: on the device; under the CPU replay of the GPU suite the same scripts run on the kernel bodies.
"jit" (round 6): random STRUCTURED robots through their run-time instantiations against the general kernels, bit for bit (device only)."""
import os
import runpy
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("family", ["dyn", "ik", "kin", "rne", "paths", "fleet", "jit"])
def test_gpu_fuzz_against_the_oracle(family, capsys):
    argv = sys.argv
    sys.argv = ["gpu_fuzz_%s.py" % family]
    try:
        with pytest.raises(SystemExit) as e:
            runpy.run_path(os.path.join(ROOT, "scripts", "gpu_fuzz_%s.py" % family), run_name="__main__")
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    assert e.value.code == 0, out[-2000:]

"""The `-m gpu` suite's HOST-buffer tests, run where no GPU exists: a child pytest with RTBHIP_TEST_CPU_REPLAY=1 (tests/conftest.py installs
tests/cpu_backend.py: the product's entry-point validation, then the kernel bodies replayed on the CPU) over every GPU test that is not listed
in tests/replay_needs_device.txt -- the handful that are ABOUT the device: graph capture, device / pinned memory accounting, the loaded
library itself, process groups on GPUs.  Device-tensor calls are replayed too: under the replay host memory wears the device label
(tests/conftest.py), so the host layer's torch branch runs here as well.  What this buys: a change to the Python layer that would break the GPU run (shapes, keywords, error types, the reference-class
and reference-suite tests) fails HERE, in the `-m "not gpu"` run, and not at the next visit to a GPU.  It is NOT the GPU run: the launch
code, the staging pipeline and the device arithmetic (same source, different compiler) are only exercised there."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIST = os.path.join(ROOT, "tests", "replay_needs_device.txt")


def test_host_buffer_gpu_tests_pass_on_the_cpu_replay():
    skip = [l.strip() for l in open(LIST) if l.strip() and not l.startswith("#")]
    cmd = [sys.executable, "-m", "pytest", "tests", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-rf"]      # no -x: every failure is shown
    for s in skip:
        cmd += ["--deselect", s]
    env = dict(os.environ, RTBHIP_TEST_CPU_REPLAY="1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().split("\n")[-40:])
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", tail)
    assert m and int(m.group(1)) >= 170, tail

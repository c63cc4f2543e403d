"""The reference's OWN unit-test files against the HOST layer of this backend where no GPU exists.

tests/test_reference_suite.py runs tests/test_ET.py ... test_Robot.py of robotics-toolbox-python, unmodified, on the MI355X.  This file runs the same
files, with the same ledger, on tests/cpu_backend.py: the product's own argument validation (the real entry points of api.cpp) followed by the
kernels' __host__ __device__ bodies replayed on the CPU.  What it covers without a GPU is everything ABOVE the C ABI -- the Python mirror of the
reference's classes: call shapes, keywords, error behaviour, the values through the kernel arithmetic -- so a host-layer regression shows up in the
`-m "not gpu"` run instead of at the next GPU visit.  Needs /root/reference (or the byte-compiled copies under oracle/_ref/pytests)."""
import pytest

import cpu_backend
import test_reference_suite as S
from oracle import ref_classes

pytestmark = pytest.mark.skipif(not ref_classes.tests_available(), reason="needs the reference's test files (oracle/_ref/pytests)")


@pytest.mark.parametrize("name", S.FILES)
def test_reference_test_file_on_the_cpu_replay(name):
    with cpu_backend.installed() as be:
        before = sum(be.calls.values())
        S.check_file(name)
        assert sum(be.calls.values()) > before, "no compute call reached the replay: the file did not run on this backend"

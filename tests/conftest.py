import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robotics-toolbox-python_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def _ensure_library():
    """The built library travels with the tree; if a checkout arrives without it (it is git-ignored) and hipcc
    is present, build it once instead of failing every test on import."""
    lib = os.path.join(PKG, "lib", "librtbhip.so")
    if not os.path.exists(lib):
        try:
            import __graft_entry__ as g
            g.build_lib()
        except Exception as e:                      # the tests that need it will say so loudly
            sys.stderr.write("conftest: could not build librtbhip.so: %r\n" % (e,))


def pytest_configure(config):
    _ensure_library()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import rtbhip
        return rtbhip.device_count() > 0
    except Exception:
        return False


def _cpu_replay():
    """DEVELOPER SWITCH, off unless RTBHIP_TEST_CPU_REPLAY=1: serve the `-m gpu` tests' HOST-buffer calls from tests/cpu_backend.py (the real
    entry points' validation + the kernel bodies replayed on the CPU) so that host-layer changes can be tried where no GPU exists.  Tests that
    need device buffers, streams or timings fail under it -- it is a way to iterate, not a substitute for the GPU run, and nothing the driver
    runs sets the variable."""
    return os.environ.get("RTBHIP_TEST_CPU_REPLAY") == "1"


@pytest.fixture(scope="session", autouse=True)
def _cpu_replay_session():
    if not _cpu_replay():
        yield
        return
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_backend
    with cpu_backend.installed():
        yield


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip silently.
    if config.getoption("-m") and "not gpu" not in config.getoption("-m") and "gpu" in config.getoption("-m"):
        if not _has_gpu() and not _cpu_replay():
            raise pytest.UsageError("-m gpu requested but librtbhip sees no HIP device (or is not built)")

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


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip silently.
    if config.getoption("-m") and "not gpu" not in config.getoption("-m") and "gpu" in config.getoption("-m"):
        if not _has_gpu():
            raise pytest.UsageError("-m gpu requested but librtbhip sees no HIP device (or is not built)")

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
    with cpu_backend.installed() as be, _host_memory_wearing_the_device_label():
        be.device_label_is_host = True
        try:
            yield
        finally:
            be.device_label_is_host = False


import contextlib


@contextlib.contextmanager
def _host_memory_wearing_the_device_label():
    """REPLAY ONLY.  The host layer has a second branch for device tensors (results allocated with torch on the caller's device, the
    caller's stream, RTBHIP_MEM_DEVICE) and that branch is host code like the rest -- round 3 lost its GPU gate to a change in it that no
    GPU-less run could see.  Under the replay `x.cuda()` therefore hands back the same CPU tensor and every tensor answers `is_cuda` with
    True, streams are no-ops, and tests/cpu_backend.py serves RTBHIP_MEM_DEVICE calls from the host memory behind those tensors.  What this
    cannot show is anything about real device memory, streams, graphs or timings: the tests that are about those stay in
    tests/replay_needs_device.txt."""
    import torch

    class _Stream:
        cuda_stream = 0

        def __init__(self, *a, **k):
            pass

        def synchronize(self):
            pass

        def wait_stream(self, other):
            pass

    saved = {"is_cuda": torch.Tensor.__dict__.get("is_cuda"), "cuda": torch.Tensor.__dict__.get("cuda")}
    saved_cuda = {k: getattr(torch.cuda, k) for k in ("synchronize", "current_stream", "Stream", "stream")}
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.Stream = _Stream
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    try:
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                delattr(torch.Tensor, k)
            else:
                setattr(torch.Tensor, k, v)
        for k, v in saved_cuda.items():
            setattr(torch.cuda, k, v)


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip silently.
    if config.getoption("-m") and "not gpu" not in config.getoption("-m") and "gpu" in config.getoption("-m"):
        if not _has_gpu() and not _cpu_replay():
            raise pytest.UsageError("-m gpu requested but librtbhip sees no HIP device (or is not built)")

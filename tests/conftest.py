import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    return os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK | os.W_OK)


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a GPU skips the gpu-marked tests instead of failing them one by
    one in rtuf_create (the driver's own runs select with -m gpu / -m "not gpu" and are unaffected)."""
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="needs a GPU (/dev/kfd is absent): run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_paged_in = set()


def page_in_rccl():
    """Reads the RCCL libraries once, front to back, before a test starts a process that links them.

    librccl.so is 0.3-0.6 GB of code objects for every GPU target; on a fresh box its first use faults it in page by page from
    the image store, and `ncclCommInitAll` (which loads the kernels of this GPU) has been seen to take more than ten minutes
    that way on some boxes and five seconds on others -- a test that starts RCCL then fails on its time-out, or appears to
    hang, for reasons that have nothing to do with the code under test.  One sequential read (read-ahead: seconds) takes that
    out of the picture; the tests' own time-outs stay as they are."""
    candidates = ["/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"]
    try:
        import torch
        candidates.append(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
    except Exception:
        pass
    for path in candidates:
        real = os.path.realpath(path)
        if real in _paged_in or not os.path.isfile(real):
            continue
        _paged_in.add(real)
        try:
            with open(real, "rb", buffering=0) as f:
                while f.read(16 << 20):
                    pass
        except OSError:
            pass

import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # On a GPU box: if anything calls abort() -- the HSA runtime does, on a thread of its own, when a kernel faults --
    # the native backtrace of that thread goes to the REAL stderr (fd capture is suspended while plugins configure,
    # so fd 2 is still the terminal here), next to faulthandler's Python frames.
    if os.environ.get("GSAGE_DEBUG_ABORT_TRACE", "1") == "1":
        import torch
        if torch.cuda.is_available():
            pkg()._native.lib().gsage_debug_abort_trace(os.dup(2))


@pytest.fixture(autouse=True)
def _settle_gpu(request):
    """After every -m gpu test: wait for the device and collect garbage.  A fault of work a test left in flight is
    then reported inside THAT test, and engines (command lists, graphs, events, side streams) are torn down while
    the device is idle instead of at whatever later allocation trips the cyclic collector."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
        gc.collect()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def pkg():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("pytorch-graphsage_amd")


@pytest.fixture(scope="session")
def gs():
    return pkg()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A new handle solves problems that fit on chip in ONE launch (csrc/small_solve.hpp), and nearly every parity test is
    # that small: left on, the streamed sweep kernels -- the hot path at BASELINE.json's sizes -- would lose the solve-level
    # coverage they have had since round 1.  So the suite keeps solves on the streamed kernels (read when a handle is
    # created), and tests/test_gpu_onchip_solve.py turns the one-launch path back on for its own cases; bench.py and
    # __graft_entry__.smoke() run with the product's default.
    os.environ.setdefault("CDH_SMALL_PATH", "0")


def _have_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should skip, not fault.
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no /dev/kfd: GPU tests run on the MI355X box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

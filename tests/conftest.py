import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_cmdline_main(config):
    """CPU runs (no GPU in the container): spread the suite over 4 pytest-xdist workers -- 12 min -> under 5 -- unless the caller chose a
    worker count (-n ..) or GT_TEST_SERIAL=1.  Never on a GPU box: the GPU tests time kernels and need their workgroups co-resident.
    The emulator library is brought up to date HERE, before the workers start, so that no two of them build it at once."""
    opt = config.option
    if hasattr(config, "workerinput") or os.environ.get("GT_TEST_XDIST_CHILD") == "1":     # (a worker runs this hook too)
        return None
    if os.environ.get("GT_TEST_SERIAL") == "1" or getattr(opt, "numprocesses", None) or not hasattr(opt, "numprocesses"):
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False) or _has_gpu():
        return None
    try:
        import xdist  # noqa: F401
        from harness import emu_lib
        emu_lib()
    except Exception:
        return None
    # (xdist's own hook ran first -- tryfirst -- and found no worker count: set what it would have set)
    os.environ["GT_TEST_XDIST_CHILD"] = "1"             # (inherited by the workers)
    opt.numprocesses = 4
    opt.dist = "load"
    opt.tx = ["popen"] * 4
    return None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)

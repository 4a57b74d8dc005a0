import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _bound_host_threads():
    """The oracle legs run torch on the host.  A GPU box shows every core of the machine but grants a share
    of them (16 for one GPU): torch's default of one thread per visible core oversubscribes that share and
    stalls (the same rule as bench.py's cpu_baseline leg)."""
    try:
        visible = len(os.sched_getaffinity(0))
    except AttributeError:
        visible = os.cpu_count() or 1
    want = str(max(1, min(16, visible)))
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ.setdefault(var, want)
    import torch
    torch.set_num_threads(int(want))
    return int(want)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config._sc_host_threads = _bound_host_threads()


def pytest_report_header(config):
    return f"host threads for the oracle legs: {getattr(config, '_sc_host_threads', '?')}"


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def _oracle_threads():
    """The CPU oracle (torch ops on <= 100 000-point chunks, nerf.py's eval_batch_size) runs fastest on ~16 threads: bench.py's
    thread sweep on the 256-thread GPU box has 16 threads ahead of 32 / 64 / 256, and torch's default there is all 256."""
    import torch
    n = os.cpu_count() or 8
    if n > 16:
        torch.set_num_threads(16)


def pytest_configure(config):
    _oracle_threads()
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs the read-only reference tree (build container only)")


def pytest_collection_modifyitems(config, items):
    from oracle.ref_shim import reference_available
    if reference_available():
        return
    skip = pytest.mark.skip(reason="reference tree not present (GPU box)")
    for item in items:
        if "needs_reference" in item.keywords:
            item.add_marker(skip)

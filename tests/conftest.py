import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref (the compiled reference; build container only)")


def pytest_collection_modifyitems(config, items):
    """Plain `pytest` on a machine without a GPU: skip the gpu tier instead of erroring in its fixtures."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible: the gpu tier runs on the MI355X box (-m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

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
    config.addinivalue_line("markers", "icamd_gpu_missing: internal -- gpu tier requested but no GPU usable (fails the test)")


def _gpu_tier_requested(config):
    """True when the run explicitly asks for the gpu tier: `-m gpu` (any mark expression that selects gpu rather than
    excluding it) or ICAMD_REQUIRE_GPU=1."""
    if os.environ.get("ICAMD_REQUIRE_GPU") == "1":
        return True
    expr = (config.getoption("markexpr", "") or "").replace("(", " ").replace(")", " ").split()
    return any(tok == "gpu" and (i == 0 or expr[i - 1] != "not") for i, tok in enumerate(expr))


def pytest_collection_modifyitems(config, items):
    """Plain `pytest` on a machine without a GPU skips the gpu tier instead of erroring in its fixtures.  When the gpu
    tier was asked for explicitly (`-m gpu` on the GPU box, or ICAMD_REQUIRE_GPU=1) a missing GPU / broken torch is a
    FAILURE of every gpu test, never a green all-skipped run."""
    why = None
    try:
        import torch
        if not torch.cuda.is_available():
            why = "torch.cuda.is_available() is False"
    except Exception as e:  # a broken torch / ROCm install
        why = "import torch failed: %r" % (e,)
    if why is None:
        return
    required = _gpu_tier_requested(config)
    skip = pytest.mark.skip(reason="no GPU visible (%s): the gpu tier runs on the MI355X box (-m gpu)" % why)
    for item in items:
        if "gpu" in item.keywords:
            if required:
                item.add_marker(pytest.mark.icamd_gpu_missing(why))
            else:
                item.add_marker(skip)


@pytest.fixture(autouse=True)
def _fail_when_required_gpu_is_missing(request):
    m = request.node.get_closest_marker("icamd_gpu_missing")
    if m is not None:
        pytest.fail("the gpu tier was requested but no GPU is usable: %s" % m.args[0], pytrace=False)

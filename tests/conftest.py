import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "fallback: exercises tmdiff_amd.fallback (kernels no BASELINE configuration reaches); "
                                       "these tests are gpu tests too and run in the same session")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must fail loudly on a box without a GPU rather than be skipped silently;
    # plain runs (no -m) on a GPU-less box skip them.
    if config.getoption("-m"):
        return
    if not torch.cuda.is_available():
        skip = pytest.mark.skip(reason="no GPU")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return load


def rel_err(a, b):
    """max|a-b| / max|b| and relative L2 error."""
    a, b = (t.detach().cpu() if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t)) for t in (a, b))
    a, b = a.double(), b.double()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / scale), float((a - b).norm() / b.norm().clamp_min(1e-30))


def assert_close(a, b, max_rel=1e-5, l2_rel=1e-5, what=""):
    m, l2 = rel_err(a, b)
    assert m <= max_rel and l2 <= l2_rel, f"{what}: max-rel {m:.3e} (<= {max_rel}), rel-L2 {l2:.3e} (<= {l2_rel})"

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


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing them
    (device_count() does not initialise the runtime)"""
    import torch
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no MI355X visible (gpu-marked test)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def unit_rows(n, dim, seed, kind="gauss"):
    """Synthetic corpora/queries. gauss: iid N(0,1) rows L2-normalised (adversarial near-ties);
    clustered: 64 centroids + noise (text-embedding-like, forces the exact fallback)."""
    rng = np.random.default_rng(seed)
    if kind == "clustered":
        cent = rng.standard_normal((64, dim)).astype(np.float32)
        x = cent[rng.integers(0, 64, n)] + 0.05 * rng.standard_normal((n, dim)).astype(np.float32)
    else:
        x = rng.standard_normal((n, dim)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return np.ascontiguousarray(x, dtype=np.float32)


def icd_levels(n, seed):
    """levels drawn from the real CSV histogram 12.43 % / 29.91 % / 57.66 %."""
    r = np.random.default_rng(seed).random(n)
    return np.where(r < 0.1243, 1, np.where(r < 0.4234, 2, 3)).astype(np.int32)


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc  # noqa
    orc.build()
    return orc


@pytest.fixture(scope="session")
def confidence_oracle():
    """oracle/confidence_oracle.py (row N3), imported the way `oracle` is: as a top-level module from oracle/"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import confidence_oracle as co  # noqa
    return co

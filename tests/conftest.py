import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "intel-texture-works-plugin_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (test infrastructure; never the product)."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def golden_inputs():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz")))


@pytest.fixture(scope="session")
def golden_blocks():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_blocks.npz")))


@pytest.fixture(scope="session")
def itw():
    """The product binding.  On a GPU box the library MUST be present and loaded (no silent fallback)."""
    import itw_amd
    itw_amd.lib()
    return itw_amd


@pytest.fixture(scope="session")
def gpu(itw):
    import torch
    assert torch.cuda.is_available(), "gpu-marked test running without a GPU"
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def first_mismatch(a, b, bpb):
    """Index of the first differing block and a short hex dump, for assertion messages."""
    a = np.asarray(a, dtype=np.uint8).reshape(-1, bpb)
    b = np.asarray(b, dtype=np.uint8).reshape(-1, bpb)
    bad = np.nonzero((a != b).any(axis=1))[0]
    if bad.size == 0:
        return None
    i = int(bad[0])
    return f"{bad.size}/{a.shape[0]} blocks differ; first #{i}: got {a[i].tobytes().hex()} want {b[i].tobytes().hex()}"

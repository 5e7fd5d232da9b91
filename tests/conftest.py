import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU: they are skipped only
    when deselected by -m; if selected without a device they FAIL in the fixture below."""


@pytest.fixture(scope="session")
def gpu():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test selected but no HIP device is visible"
    import umeregrobust_amd as ume
    ume.require_native()
    return torch.device("cuda:0")

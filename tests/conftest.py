import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True, scope="session")
def _check_early_flags():
    """Every backward pass of the suite that takes the early-flags path (GradCoordinator.offer_early) also reads the
    flags its weight-gradient launches raise and fails on any difference."""
    from ironcortex_amd.hip import ops

    ops.GradCoordinator.check_early = True
    yield

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


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


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_big_device_buffers():
    """The full-size tests hold ~200 GB (stored pre-activations of the 8.2 M-pair grid); hand the blocks back to the
    driver afterwards so the next full-size test does not have to fit beside torch's cached, differently sized ones."""
    yield
    import torch

    if torch.cuda.is_available() and torch.cuda.memory_reserved() > 32e9:
        import gc

        import protnote_amd

        gc.collect()
        protnote_amd.free_workspaces()
        torch.cuda.empty_cache()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU oracle runs (set VP_SLOW=1 to enable)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("VP_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow oracle run; set VP_SLOW=1")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_rows():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "survey_table.json")) as f:
        return json.load(f)["rows"]


@pytest.fixture(scope="session")
def engine():
    """One Engine per test session.  Fails (does not skip) when the GPU or libvphip.so is missing."""
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a GPU"
    from cuda_mesh_voxelization_amd.pipeline import Engine
    return Engine(0)

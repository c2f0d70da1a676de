import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)



@pytest.fixture
def dummy_objectives(monkeypatch):
    """Objectives built from placeholders (None, strings) for structural tests: switch the constructor's
    validation off, as scripts do with ``krotov.Objective.type_checking = False``."""
    import krotov_amd

    monkeypatch.setattr(krotov_amd.Objective, 'type_checking', False)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "no_oracle: a GPU test that compares with no oracle / reference data (timing, "
                                       "plumbing): its kernel launches do not count for tests/test_zz_kernel_coverage.py")
    # kernel-instantiation coverage (include/krotov_hip.h: kh_debug_launched): one log per session; the library appends
    # an instantiation's name at its first launch in a process while KH_LAUNCH_LOG names the file -- which the fixture
    # below arranges for the oracle-comparing GPU tests only (their rank sub-processes inherit the variable)
    import tempfile

    fd, path = tempfile.mkstemp(prefix='kh_launch_log_', suffix='.txt')
    os.close(fd)
    config._kh_launch_log = path
    # (a full run in ONE process: pytest-xdist workers each see a part of the suite and keep their own log)
    config._kh_full_run = (not config.getoption('keyword') and all(os.path.isdir(a.split('::')[0]) for a in config.args)
                           and not hasattr(config, 'workerinput') and not getattr(config.option, 'numprocesses', None))


def pytest_unconfigure(config):
    path = getattr(config, '_kh_launch_log', None)
    if path and os.path.exists(path):
        os.unlink(path)


@pytest.fixture(autouse=True)
def _kernel_launch_log(request, monkeypatch):
    if 'gpu' in request.keywords and 'no_oracle' not in request.keywords:
        monkeypatch.setenv('KH_LAUNCH_LOG', request.config._kh_launch_log)
    else:
        monkeypatch.delenv('KH_LAUNCH_LOG', raising=False)
    yield


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

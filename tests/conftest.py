import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_collection_finish(session):
    """Start the CPU-oracle process pool (tests/oracle_jobs.py) for the large parity scenes the selected tests will ask for, so
    that their fp32 / fp64 oracle evaluations run on the host cores WHILE the GPU tests run."""
    wanted = []
    for item in session.items:
        if item.get_closest_marker("gpu") is None:
            continue
        try:
            import oracle_jobs
        except Exception:
            return
        for pat, keys in oracle_jobs.WANTS.items():
            if pat in item.nodeid:
                wanted += keys
    if not wanted or session.config.option.collectonly:
        return
    import torch
    if torch.cuda.is_available():
        import oracle_jobs
        oracle_jobs.start(wanted)


def pytest_sessionfinish(session, exitstatus):
    mod = sys.modules.get("oracle_jobs")
    if mod is not None:
        mod.shutdown()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _restore_library_switches():
    """The C ABI's process-wide schedule switches (sdumc_set_concurrency / _background_lane / _chain_cluster / sdumc_set_split_) are restored to
    their defaults after EVERY test, so that a test that flips one and then fails cannot change the schedule the rest of the
    session runs (tests still restore them themselves in try/finally; this is the backstop)."""
    yield
    mod = sys.modules.get("sdumc_amd._lib")
    if mod is None:
        return
    lib = mod.lib
    lib.sdumc_set_concurrency(1)
    lib.sdumc_set_background_lane(3)
    e = os.environ.get("SDUMC_CHAIN_CLUSTER")
    lib.sdumc_set_chain_cluster(int(e) if e else 1)
    e = os.environ.get("SDUMC_SPLIT")
    lib.sdumc_set_split_(int(e) if e else 15)

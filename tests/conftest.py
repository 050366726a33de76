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
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def gapped4000():
    from eigensolvers_amd.generators import gapped_csr_host, guess_vector
    return gapped_csr_host(4000, 32, seed=7), guess_vector(4000, 1)


@pytest.fixture(scope="session")
def hip():
    """The product package with a live device context; GPU tests only."""
    import eigensolvers_amd as ea
    ea.HipContext.default()          # raises loudly when the library or the GPU is missing
    return ea

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:                 # helper modules of the tests (mask_cases)
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hm():
    import hydra_mi
    return hydra_mi


@pytest.fixture(scope="session")
def oracle_brox():
    from oracle import brox_oracle
    brox_oracle.build()
    return brox_oracle

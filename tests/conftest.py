import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    return GOLD


@pytest.fixture(scope="session")
def sd24():
    from targetdiarization_amd.weights import recipe_state_dict
    return recipe_state_dict(seed=0, num_blocks=24)


@pytest.fixture(scope="session")
def sd2():
    from targetdiarization_amd.weights import recipe_state_dict
    return recipe_state_dict(seed=1, num_blocks=2)

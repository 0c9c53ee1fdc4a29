import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = os.path.join(ROOT, "tests", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run on the GPU box with -m gpu)")


def scene_path(name):
    return os.path.join(SCENES, name + ".obj")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_py

    return oracle_py.load()

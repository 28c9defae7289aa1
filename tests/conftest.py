import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)          # tests/scenes.py, tests/gpu_util.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pt_oracle_py as po
    po.lib()
    return po


@pytest.fixture(scope="session")
def pt():
    import pathtracer_amd
    pathtracer_amd.load_library()
    return pathtracer_amd

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    from deflatedmlmc_schwinger_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return engine.load_library()

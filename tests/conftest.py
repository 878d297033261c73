import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
# the file mapper (mp_map_file) takes every GPU of the node by default; the tests are written for one (tests/test_multi_gpu.py names its own list)
os.environ.setdefault("MPA_DEVICES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    import refbind
    refbind.build_oracle()
    return True

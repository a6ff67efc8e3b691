import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU restatement (oracle/liboracle.so); built on demand."""
    import orc
    return orc.load()


@pytest.fixture(scope="session")
def workdir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("bmbs"))


def ref_binary():
    p = os.path.join(ROOT, "oracle", "_ref", "bitmapperBS")
    return p if os.path.exists(p) else None

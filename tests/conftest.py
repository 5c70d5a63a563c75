import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Builds (if hipcc is here and sources are newer) and loads the C-ABI library."""
    from mecano_amd import build, _lib
    try:
        build.build_lib()
    except Exception:
        if not os.path.exists(build.LIB):
            raise
    return _lib.load()

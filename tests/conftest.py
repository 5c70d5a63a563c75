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
    """Builds (if hipcc is here and the library does not carry the build id of the current sources) and loads the C-ABI library."""
    from mecano_amd import build, _lib
    try:
        build.build_lib()
    except Exception:
        if not os.path.exists(build.LIB):
            raise
    return _lib.load()


def pytest_sessionfinish(session, exitstatus):
    """The achieved parity errors of the run (tests/helpers.py: PARITY_LOG), for the record under profiles/."""
    try:
        import json
        from helpers import PARITY_LOG
        if PARITY_LOG:
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "parity_errors.json"), "w") as f:
                json.dump({"exitstatus": int(exitstatus), "tests": PARITY_LOG}, f, indent=1, sort_keys=True)
    except Exception:
        pass

"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the CPU build only; GPU ASan is
not available on this pool).  oracle/sanitize_driver.c includes the oracle's source and calls every entry point on a 24-joint tree with
every joint kind, locked joints, offset centres of mass and external wrenches; any finding aborts it with a non-zero exit code."""
import os
import shutil
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.mark.skipif(shutil.which("gcc") is None and shutil.which("cc") is None, reason="no C compiler")
def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", ORACLE, "-B", "sanitize_driver"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([os.path.join(ORACLE, "sanitize_driver")], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "sanitize_driver:" in run.stdout and "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr

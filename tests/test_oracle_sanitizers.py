"""CPU: the C oracle under AddressSanitizer and UndefinedBehaviorSanitizer.  The oracle is the
checker of every GPU result, so a memory error or an overflow inside it would be a wrong answer
that nothing else looks at.  oracle/sanitize_driver.c drives every entry point (60,000 positions
of random play through both move generators, six self-play games with noise, tree reuse, the
uniform-prior fallback and a restart); the sanitizer runtimes are linked statically into that
executable, nothing is preloaded."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE = os.path.join(os.path.dirname(HERE), "oracle")


@pytest.mark.skipif(shutil.which("gcc") is None or shutil.which("make") is None, reason="needs gcc and make")
def test_oracle_is_clean_under_asan_and_ubsan():
    b = subprocess.run(["make", "-C", ORACLE, "-B", "sanitize"], capture_output=True, text=True, timeout=300)
    if b.returncode != 0 and ("libasan" in b.stderr or "libubsan" in b.stderr):
        pytest.skip("this gcc has no static sanitizer runtimes")
    assert b.returncode == 0, b.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ORACLE, "_san", "sanitize_driver")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "sanitize_driver ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr

"""The reference's own CPU test files (game rules, encodings, network module, baseline
cross-check, trainer checkpoint logic) run unchanged against the drop-in modules.
Skipped where the reference checkout is absent (it never travels to the GPU box)."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ["tests/test_game.py", "tests/test_model.py", "tests/test_compatibility.py", "tests/test_trainer.py"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="reference checkout not present")
def test_reference_cpu_tests_pass_against_the_dropins(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "run_reference_tests.py"), REF] + FILES,
                       cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout

"""Runs the reference's OWN test files against the drop-in modules: `src.game`, `src.mcts`,
`src.model`, `src.config` and `src.parallel_trainer` are replaced by their alphazero_amd
counterparts before the tests import them (everything else, e.g. `src.baseline`, stays the
reference's).  Only usable where the reference checkout exists:

    python tests/run_reference_tests.py /root/reference tests/test_game.py tests/test_model.py ...
"""
import os
import sys

ref = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ref)
sys.dont_write_bytecode = True          # leave the reference checkout untouched

import src  # noqa: E402  (the reference's package shell)
import alphazero_amd.config  # noqa: E402
import alphazero_amd.game  # noqa: E402
import alphazero_amd.mcts  # noqa: E402
import alphazero_amd.model  # noqa: E402
import alphazero_amd.parallel_trainer  # noqa: E402

for name in ("game", "mcts", "model", "config", "parallel_trainer"):
    mod = getattr(alphazero_amd, name)
    sys.modules["src." + name] = mod
    setattr(src, name, mod)

import pytest  # noqa: E402

sys.exit(pytest.main(["-q", "-x", "-p", "no:cacheprovider", "--rootdir=/tmp"]
                     + [os.path.join(ref, f) for f in sys.argv[2:]]))

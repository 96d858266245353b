"""GPU: a few rounds of the randomized device-against-oracle comparison
(tests/fuzz_tree_parity.py holds the scenario generator)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_scenarios_match_the_oracle(seed):
    from fuzz_tree_parity import one_round
    assert one_round(seed, n_games=12, max_plies=6, max_sims=80) > 0

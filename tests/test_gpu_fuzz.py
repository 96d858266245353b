"""GPU: a few rounds of the randomized device-against-oracle comparison
(tests/fuzz_tree_parity.py holds the scenario generator)."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_scenarios_match_the_oracle(seed):
    from fuzz_tree_parity import one_round
    assert one_round(seed, n_games=12, max_plies=6, max_sims=80) > 0


@pytest.mark.parametrize("seed", [21, 22, 23, 24, 25, 26])
def test_random_fused_searches_match_the_step_api(seed):
    from fuzz_fused_parity import one_round
    assert one_round(seed, sizes=(3, 37, 130, 300, 520), max_sims=24, max_plies=2) >= 0


@pytest.mark.parametrize("seed", [31, 32, 33, 34, 35, 36])
def test_random_networks_and_batches_stay_within_tolerance(seed):
    from fuzz_net_parity import one_round
    err_p, err_v, _, _, _ = one_round(seed, max_batch=700)
    assert err_p <= 2e-2 and err_v <= 2e-2

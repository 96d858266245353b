"""GPU: statistical parity of the two device-side random draws of fused self-play with
what the reference draws on the host — np.random.dirichlet([0.35] * k) mixed into the
root priors (src/mcts.py:314-326) and np.random.choice(192, p=probs) for the move
(src/parallel_trainer.py:189).  Seeds are fixed, so the statistics below are
deterministic numbers, not flaky samples."""
import numpy as np
import pytest
import torch
from scipy import stats

pytestmark = pytest.mark.gpu

ALPHA, EPS = 0.35, 0.25


def _noisy_root_priors(n_games, seed):
    """Start position in every game, uniform network output, device-drawn noise:
    returns (k, P0 float32[k], P float64[n_games, k])."""
    from alphazero_amd.engine import Engine
    e = Engine(n_games, pool_nodes_per_game=256, seed=seed)
    e.reset_all()
    e.set_active(np.ones(n_games, np.uint8))
    e.roots_prepare()
    rs = np.random.RandomState(3)
    probs = rs.rand(192).astype(np.float32)
    probs /= probs.sum()
    e.roots_expand(torch.from_numpy(np.tile(probs, (n_games, 1))).cuda())
    before = e.roots_children(np.arange(n_games))
    e.add_noise(None)
    after = e.roots_children(np.arange(n_games))
    k = len(before[0]["actions"])
    p0 = before[0]["P"].astype(np.float32)
    assert all(len(q["actions"]) == k and np.all(q["p_is64"] == 1) for q in after)
    return k, p0, np.stack([q["P"] for q in after])


def test_device_dirichlet_noise_is_dirichlet():
    n = 4096
    k, p0, P = _noisy_root_priors(n, seed=11)
    assert k == 22
    # P = double(float32(0.75f * P0)) + 0.25 * noise   (the reference's mixed-precision order)
    kept = (np.float32(1.0 - EPS) * p0).astype(np.float64)
    noise = (P - kept[None, :]) / EPS
    assert noise.min() >= -1e-12
    assert np.allclose(noise.sum(1), 1.0, atol=1e-9)
    a0 = ALPHA * k
    mean, var = 1.0 / k, (1.0 / k) * (1.0 - 1.0 / k) / (a0 + 1.0)
    # component means: standard error sqrt(var / n); 5 sigma band for each of the 22 components
    assert np.all(np.abs(noise.mean(0) - mean) < 5.0 * np.sqrt(var / n))
    assert abs(noise.var() / var - 1.0) < 0.05
    # a component of Dirichlet(alpha, ..., alpha) is Beta(alpha, (k - 1) alpha): Kolmogorov-Smirnov
    for comp in (0, 7, 21):
        d, _ = stats.kstest(noise[:, comp], stats.beta(ALPHA, ALPHA * (k - 1)).cdf)
        assert d < 1.95 / np.sqrt(n), (comp, d)           # 0.1 % critical value
    # components of one draw are negatively correlated: corr = -1 / (k - 1)
    c = np.corrcoef(noise[:, 0], noise[:, 1])[0, 1]
    assert abs(c - (-1.0 / (k - 1))) < 0.06


def test_device_noise_depends_on_seed_and_game_only():
    _, _, a = _noisy_root_priors(64, seed=5)
    _, _, b = _noisy_root_priors(64, seed=5)
    _, _, c = _noisy_root_priors(64, seed=6)
    assert np.array_equal(a, b)                            # reproducible
    assert not np.array_equal(a, c)                        # seed matters
    assert len({tuple(row) for row in a}) == 64            # every game draws its own vector


def test_device_move_choice_follows_the_visit_distribution():
    """Noise off and identical games: every game has the same visit distribution pi after
    the search; the moves drawn on the device must be distributed as pi (chi-square)."""
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet, HipInference
    n, S = 4096, 64
    torch.manual_seed(5)
    inf = HipInference(AlphaZeroNet(num_blocks=1, num_filters=128).eval(), max_batch=n)
    e = Engine(n, pool_nodes_per_game=1 << 12, seed=21)
    e.attach_net(inf)
    e.reset_all()
    e.set_active(np.ones(n, np.uint8))
    e.search(S, add_noise=False)
    root = e.root_children(0)
    pi = np.zeros(192)
    pi[root["actions"]] = root["N"] / root["N"].sum()      # temperature 1 (ply 0 < 16)
    e.play_move(0, add_noise=False, restart_finished=False)   # no further simulations: record + choose + step
    chosen = e.chosen_actions()
    assert np.all(pi[chosen] > 0)                          # only visited moves are ever played
    support = np.flatnonzero(pi > 0)
    observed = np.array([(chosen == a).sum() for a in support], np.float64)
    expected = pi[support] * n
    keep = expected >= 5                                   # chi-square validity
    chi2 = ((observed[keep] - expected[keep]) ** 2 / expected[keep]).sum()
    assert chi2 < stats.chi2(int(keep.sum()) - 1).ppf(0.999), (chi2, int(keep.sum()))

"""GPU: the ParallelTrainer drop-in end to end — the host-loop self-play against the
reference's recorded (state, policy, wl) stream, and the fused device self-play ->
replay buffer -> training step -> refreshed inference cycle."""
import os

import numpy as np
import pytest
import torch

from test_gpu_dropin import _table_mcts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture", ["selfplay_tier.npz", "selfplay_wide_tier.npz"])
def test_host_loop_reproduces_reference_examples(golden_dir, tmp_path, monkeypatch, fixture):
    """execute_parallel_episodes of the reference with a table network and recorded
    numpy draws (tests/golden/make_golden.py selfplay): same examples, same order."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ParallelTrainer
    d = np.load(os.path.join(golden_dir, fixture))
    n_games, sims = int(d["n_games"]), int(d["sims"])
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=8), device="cuda", num_parallel_games=n_games,
                         num_simulations=sims, checkpoint_dir=str(tmp_path), verbose=False)
    tr.mcts = _table_mcts(sims)
    draws = {"dir": 0, "choice": 0}

    def replay_dirichlet(alpha, *a, **k):
        v = d["dirichlet"][draws["dir"]][:len(alpha)]
        draws["dir"] += 1
        return v

    def replay_choice(n, *a, **k):
        v = int(d["choices"][draws["choice"]])
        draws["choice"] += 1
        return v
    monkeypatch.setattr(np.random, "dirichlet", replay_dirichlet)
    monkeypatch.setattr(np.random, "choice", replay_choice)
    ex = tr.execute_parallel_episodes(n_games)
    assert draws["dir"] == len(d["dirichlet"]) and draws["choice"] == len(d["choices"])
    n = int(d["n_examples"])
    assert len(ex) == n
    states = np.unpackbits(d["states"])[: n * 192].reshape(n, 3, 8, 8).astype(np.float32)
    assert np.array_equal(ex.states, states)
    assert np.array_equal(ex.policies.view(np.uint32), d["policies"].view(np.uint32))
    assert np.array_equal(ex.wls, d["wls"])


def test_fused_selfplay_train_refresh_cycle(tmp_path):
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ExampleArrays, ParallelTrainer
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=2, num_filters=128).cuda()
    tr = ParallelTrainer(model, device="cuda", num_parallel_games=8, num_simulations=16,
                         checkpoint_dir=str(tmp_path), seed=3, verbose=False)
    ex = tr.execute_parallel_episodes()
    assert tr._selfplay is not None                          # the fused engine path ran
    assert isinstance(ex, ExampleArrays) and len(ex) >= 8 * 10
    assert np.allclose(ex.policies.sum(1), 1.0, atol=1e-5)
    assert np.all(ex.states[:, 2] == 1.0) and set(np.unique(ex.states)) <= {0.0, 1.0}
    assert np.all(ex.wls.sum(1) == 1.0)
    st = tr._selfplay.engine.stats()
    assert st["games_finished"] == 8
    tr.add_examples(ex)
    tr.append_training_data(ex)
    probe = ex.states[:4]
    inf = tr._selfplay.inference
    before = inf.forward_planes(torch.from_numpy(probe).cuda())[0].clone()
    stats = tr.learn(epochs=1, batch_size=64)
    assert np.isfinite(stats["loss"]) and tr._weights_changed
    ex2 = tr.execute_parallel_episodes(4)                    # refreshes the HIP copy of the weights first
    assert not tr._weights_changed and len(ex2) >= 4 * 10
    after = inf.forward_planes(torch.from_numpy(probe).cuda())[0]
    assert not torch.equal(before, after)
    # the refreshed HIP inference agrees with the trained torch module
    model.eval()
    with torch.no_grad():
        want = torch.softmax(model(torch.from_numpy(probe).cuda())[0], 1)
    assert float((after - want).abs().max()) < 2e-2
    tr.save_iteration_checkpoint(1)
    assert tr.get_latest_iteration() == 1
    assert tr.load_training_data() == len(ex)


def test_continuous_collection_keeps_every_slot_busy(tmp_path):
    """continuous=True: the engine's slots all stay active, a call returns once the asked
    number of games has finished, games in progress carry over to the next call, and the
    examples have the same form as the episodic ones."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ParallelTrainer
    torch.manual_seed(1)
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=128).cuda(), device="cuda", num_parallel_games=16,
                         num_simulations=8, checkpoint_dir=str(tmp_path), seed=5, verbose=False)
    first = tr.execute_parallel_episodes(6, continuous=True)
    eng = tr._selfplay.engine
    assert eng.n == 16
    done1, moves1 = eng.stats()["games_finished"], eng.stats()["moves"]
    assert done1 >= 6 and len(first) >= 6 * 10
    second = tr.execute_parallel_episodes(6, continuous=True)
    st = eng.stats()
    assert st["games_finished"] - done1 >= 6 and st["moves"] > moves1
    for ex in (first, second):
        assert np.allclose(ex.policies.sum(1), 1.0, atol=1e-5)
        assert np.all(ex.wls.sum(1) == 1.0) and np.all(ex.states[:, 2] == 1.0)
    # the second call did not start from scratch: its first finished games needed fewer plies inside
    # the call than a whole game takes, so it finished its six games in fewer moves than the first
    assert st["moves"] - moves1 < moves1
    # the episodic mode afterwards starts clean
    third = tr.execute_parallel_episodes(4)
    assert len(third) >= 4 * 10

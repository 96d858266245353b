"""GPU: the ParallelTrainer drop-in end to end — the host-loop self-play against the
reference's recorded (state, policy, wl) stream, and the fused device self-play ->
replay buffer -> training step -> refreshed inference cycle."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc
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


def test_the_reference_test_network_trains_through_the_fused_path(tmp_path):
    """The network of the reference's own trainer tests (2 x 32, tests/test_trainer.py:22) on a GPU:
    execute_parallel_episodes runs through the fused device self-play (any-width kernels), learn()
    updates the weights, and the next episodes use the updated weights (the device copy is refreshed)."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ParallelTrainer
    torch.manual_seed(2)
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=2, num_filters=32).cuda(), device="cuda", num_parallel_games=8,
                         num_simulations=16, checkpoint_dir=str(tmp_path), seed=7, verbose=False)
    assert tr._fused_ok()
    ex = tr.execute_parallel_episodes(8)
    assert tr._selfplay is not None and tr._selfplay.engine.stats()["games_finished"] == 8
    assert len(ex) >= 8 * 10
    assert np.allclose(ex.policies.sum(1), 1.0, atol=1e-5)
    assert np.all(ex.wls.sum(1) == 1.0) and np.all(ex.states[:, 2] == 1.0)
    tr.add_examples(ex)
    before = [p.detach().clone() for p in tr.model.parameters()]
    stats = tr.learn(epochs=1, batch_size=32)
    assert np.isfinite(stats["loss"])
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.model.parameters()))
    assert tr._weights_changed
    again = tr.execute_parallel_episodes(4)
    assert not tr._weights_changed and len(again) >= 4 * 10


def test_learn_on_the_gpu_reproduces_the_reference_weights(tmp_path):
    """The reference trainer's learn(epochs=2, batch_size=16) on 48 synthetic examples from fixed
    seeds (tests/golden/train_tier.npz, recorded on CPU) against the drop-in's learn() on
    cuda: same shuffle (the DataLoader draws from the CPU generator), same batches, same
    losses; the arithmetic is MIOpen/rocBLAS float32 instead of the CPU's.
    Stated tolerance: every weight of the two stored tensors within 2e-4, per-tensor sums
    within 2e-4 of the tensor's absolute sum + 1e-4 (measured on MI355X: 1.7e-5, 8e-8 and
    1e-6; the slack is for Adam, whose first steps divide a gradient by its own magnitude
    and so amplify float32 noise on elements whose gradient is close to zero)."""
    from test_trainer_dropin import GOLD, train_examples
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ParallelTrainer
    torch.manual_seed(123)
    model = AlphaZeroNet(num_blocks=1, num_filters=8).cuda()
    tr = ParallelTrainer(model, device="cuda", num_parallel_games=2, num_simulations=4,
                         checkpoint_dir=str(tmp_path), verbose=False)
    tr.add_examples(train_examples())
    assert tr._replay.s.is_cuda                                     # the buffer lives on the training device
    torch.manual_seed(7)
    stats = tr.learn(epochs=2, batch_size=16)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert list(sd.keys()) == [str(k) for k in GOLD["learn_keys"]]
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    abs_sums = np.array([float(v.double().abs().sum()) for v in sd.values()])
    err_conv = np.abs(sd["input_conv.weight"].numpy() - GOLD["learn_input_conv"]).max()
    err_fc = np.abs(sd["value_fc2.weight"].numpy() - GOLD["learn_value_fc2"]).max()
    err_sums = np.abs(sums - GOLD["learn_sums"]) / (GOLD["learn_abs_sums"] + 0.5)
    print(f"gpu learn vs reference: input_conv max-abs {err_conv:.2e}, value_fc2 max-abs {err_fc:.2e}, "
          f"worst per-tensor sum error relative to its absolute sum {err_sums.max():.2e}")
    assert err_conv <= 2e-4 and err_fc <= 2e-4
    assert np.all(np.abs(sums - GOLD["learn_sums"]) <= 2e-4 * GOLD["learn_abs_sums"] + 1e-4)
    np.testing.assert_allclose(abs_sums, GOLD["learn_abs_sums"], rtol=2e-4, atol=1e-4)
    assert tr.optimizer.param_groups[0]["lr"] == pytest.approx(float(GOLD["learn_lr_after"]), rel=1e-12)
    assert np.isfinite(stats["loss"])


def test_examples_stay_on_the_device_from_the_queue_to_the_optimiser(tmp_path, monkeypatch):
    """execute_parallel_episodes -> add_examples -> learn on the fused path: the examples are
    cuda tensors all the way; the numpy view of the buffer is only made when somebody reads it."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ExampleArrays, ParallelTrainer
    torch.manual_seed(4)
    model = AlphaZeroNet(num_blocks=1, num_filters=128).cuda()
    tr = ParallelTrainer(model, device="cuda", num_parallel_games=8, num_simulations=8,
                         checkpoint_dir=str(tmp_path), seed=9, verbose=False)
    ex = tr.execute_parallel_episodes()
    assert ex._np is None and ex._t[0].is_cuda                      # nothing came to the host
    tr.add_examples(ex)
    assert tr._replay.s.is_cuda and len(tr._replay) == len(ex)
    calls = []
    monkeypatch.setattr(ExampleArrays, "_host", lambda self: calls.append(1) or (_ for _ in ()).throw(AssertionError("host copy")))
    stats = tr.learn(epochs=1, batch_size=32)                       # gathers from the ring on the device
    assert np.isfinite(stats["loss"]) and not calls
    monkeypatch.undo()
    assert tr._weights_changed                                      # version counters moved: the HIP copy is stale
    model.load_state_dict(model.state_dict())                       # any in-place change counts, not only learn()
    ex2 = tr.execute_parallel_episodes(4)
    assert not tr._weights_changed and len(ex2) >= 4 * 10
    view = tr.examples                                              # now the host view is built, once
    assert view.states.shape == (len(ex), 3, 8, 8) and np.array_equal(view.policies, ex.policies)
    assert tr.examples is view


def test_one_game_100_sims_full_network_against_the_oracle(tmp_path, monkeypatch):
    """The plumbing configuration (1 self-play game x 100 simulations per move, SE-ResNet 20 x 128,
    root noise, temperature schedule, tree reuse, played to the end) through the drop-in trainer's
    reference loop on the HIP path, against the CPU oracle driven by the same network's outputs and
    the same numpy draws: every (state, pi, wl) example equal bit for bit."""
    from selfplay_replay import replay_selfplay_oracle
    from alphazero_amd.model import AlphaZeroNet, HipInference
    from alphazero_amd.parallel_trainer import ParallelTrainer
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=20, num_filters=128).cuda().eval()
    tr = ParallelTrainer(model, device="cuda", num_parallel_games=1, num_simulations=100,
                         checkpoint_dir=str(tmp_path), verbose=False)
    dirichlet, choices = [], []
    real_dirichlet, real_choice = np.random.dirichlet, np.random.choice

    def rec_dirichlet(alpha, *a, **k):
        v = real_dirichlet(alpha, *a, **k)
        dirichlet.append(np.concatenate([v, np.zeros(48 - len(v))]))
        return v

    def rec_choice(n, *a, **k):
        v = real_choice(n, *a, **k)
        choices.append(int(v))
        return v
    monkeypatch.setattr(np.random, "dirichlet", rec_dirichlet)
    monkeypatch.setattr(np.random, "choice", rec_choice)
    np.random.seed(0)
    ex = tr._host_loop_episodes(1)                      # the reference's loop over the drop-in MCTS
    monkeypatch.undo()
    assert tr.mcts._hip is not None                     # the searches ran on the HIP network
    assert len(ex) == len(choices) >= 20 and np.all(ex.wls.sum(1) == 1.0)

    inf = HipInference(model, max_batch=16)
    start = orc.encode_state(orc.start_position())

    def evaluate(planes, need):
        x = np.ascontiguousarray(planes, np.float32).copy()
        x[np.asarray(need) == 0] = start
        probs, value, _, _ = inf.forward_planes(torch.from_numpy(x).cuda())
        return probs.cpu().numpy().copy(), value.cpu().numpy().copy()
    s, p, w = replay_selfplay_oracle(1, 100, np.array(dirichlet), np.array(choices), evaluate)
    assert np.array_equal(ex.states, np.stack(s))
    assert np.array_equal(ex.policies.view(np.uint32), np.stack(p).view(np.uint32))
    assert np.array_equal(ex.wls, np.stack(w))


def test_one_iteration_of_the_reference_training_loop(tmp_path):
    """The body of the reference's main.train loop (main.py:66-90), statement for statement, on the
    drop-in trainer with the fused device self-play: several self-play batches collected into a
    Python list and into the replay buffer, learn(), the data file, the iteration checkpoint,
    and the restart path (get_latest_iteration / load_iteration_checkpoint / load_training_data,
    main.py:54-59)."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ParallelTrainer
    torch.manual_seed(3)
    model = AlphaZeroNet(num_blocks=1, num_filters=128).to("cuda")
    trainer = ParallelTrainer(model, device="cuda", num_parallel_games=8, num_simulations=8,
                              checkpoint_dir=str(tmp_path), seed=1, verbose=False)
    all_new_examples = []
    for _ in range(2):
        new_examples = trainer.execute_parallel_episodes(num_games=8)
        all_new_examples.extend(new_examples)
        trainer.add_examples(new_examples)
    assert len(all_new_examples) == len(trainer.examples) >= 2 * 8 * 10
    state, policy, wl = all_new_examples[0]
    assert state.shape == (3, 8, 8) and policy.shape == (192,) and wl.shape == (2,)
    stats = trainer.learn(epochs=1, batch_size=64)
    assert np.isfinite(stats["loss"])
    trainer.append_training_data(all_new_examples)
    trainer.save_iteration_checkpoint(1)
    # restart
    torch.manual_seed(4)
    again = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=128).to("cuda"), device="cuda", num_parallel_games=8,
                            num_simulations=8, checkpoint_dir=str(tmp_path), seed=1, verbose=False)
    assert again.get_latest_iteration() == 1 and again.load_iteration_checkpoint(1)
    assert again.load_training_data() == len(all_new_examples)
    for a, b in zip(model.state_dict().values(), again.model.state_dict().values()):
        assert torch.equal(a, b)
    assert np.array_equal(again.examples.policies, trainer.examples.policies)
    assert len(again.execute_parallel_episodes(num_games=4)) >= 4 * 10    # and it plays on with the restored weights


def test_graphed_learn_matches_the_eager_step(tmp_path):
    """ParallelTrainer(graph_step=True): every full batch of learn() is replayed from captured HIP
    graphs (the regime of a data-parallel rank's small slice, where the step is launch bound); the
    ragged last batch and the first steps run eagerly.  Same seeds, same examples as an eager
    trainer.  The two runs cannot agree bit for bit: capturable Adam rounds differently in the
    seventh digit, MIOpen's weight-gradient kernels are not deterministic, and a pre-activation
    within that noise of zero flips its ReLU mask, which changes single gradient elements by
    percents; over tens of steps that divergence grows like that of any two float32 training runs
    (measured after 52 steps: relative L2 distance 1e-2 to 3e-2 between any two of the three runs
    below, in no fixed order).  So the arithmetic of the replayed step is checked where the
    divergence has had no time to grow: after six steps (three eager warm-up steps, the capture,
    two more replays) the graphed weights are within 1e-3 of eager steps with the same optimiser
    arithmetic - a skipped, doubled or stale step would be 1e-2 away.  The long run (two learn()
    calls with a learning-rate change between them) then has to keep the losses together, stay
    inside the divergence two eager runs show, and re-capture the update graph at the new
    learning rate; a checkpoint written by the graphed trainer loads into an eager one on the CPU."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.parallel_trainer import ExampleArrays, ParallelTrainer
    rs = np.random.RandomState(3)
    n = 200
    states = (rs.rand(n, 3, 8, 8) > 0.7).astype(np.float32)
    pol = rs.rand(n, 192).astype(np.float32)
    pol /= pol.sum(1, keepdims=True)
    wls = np.tile(np.array([[1.0, 0.0], [0.0, 1.0]], np.float32), (n // 2, 1))

    def run(graph, capturable=False, n_used=n, calls=2, epochs=2, tag=""):
        torch.manual_seed(11)
        model = AlphaZeroNet(num_blocks=2, num_filters=32).cuda()
        tr = ParallelTrainer(model, device="cuda", checkpoint_dir=str(tmp_path / (("g" if graph else "e") + tag)), verbose=False,
                             graph_step=graph)
        if capturable:
            from alphazero_amd.distributed import make_capturable
            make_capturable(tr.optimizer)
        tr.add_examples(ExampleArrays(states[:n_used], pol[:n_used], wls[:n_used]))
        out = []
        for _ in range(calls):
            torch.manual_seed(5)
            out.append(tr.learn(epochs=epochs, batch_size=32))
        return tr, out

    def distance(x, y):
        with torch.no_grad():
            px = torch.cat([t.float().reshape(-1) for t in x.model.parameters()])
            py = torch.cat([t.float().reshape(-1) for t in y.model.parameters()])
            return float((px - py).norm() / px.norm()), float((px - py).abs().max())

    # ---- six steps: 96 examples, mirrored to 192 = six full batches of 32, one epoch
    short_e, _ = run(False, capturable=True, n_used=96, calls=1, epochs=1, tag="s")
    short_g, _ = run(True, n_used=96, calls=1, epochs=1, tag="s")
    step = short_g._graphed[32]
    assert step.graph_a is not None and step.graph_b is not None and step.steps_seen == 6
    d_short = distance(short_e, short_g)
    print(f"after 6 steps (3 replayed): relative L2 distance / largest single difference {d_short[0]:.2e} / {d_short[1]:.2e}")
    assert d_short[0] < 1e-3, d_short

    # ---- 52 steps over two learn() calls: 400 augmented examples = 12 full batches + one of 16 per epoch
    eager, stats_e = run(False)
    same_adam, _ = run(False, capturable=True)        # eager steps with the optimiser arithmetic the graphs use
    graphed, stats_g = run(True)
    assert graphed._graphed and all(g.graph_a is not None for g in graphed._graphed.values())
    for a, b in zip(stats_e, stats_g):
        for k in ("loss", "pi", "wl"):
            assert abs(a[k] - b[k]) < 2e-2, (k, a[k], b[k])
    d_graph, d_adam = distance(same_adam, graphed), distance(eager, same_adam)
    print(f"after 52 steps, relative L2 distance / largest single difference of the weights: graphed against eager steps "
          f"with the same optimiser arithmetic {d_graph[0]:.2e} / {d_graph[1]:.2e}; those against the reference's Adam "
          f"{d_adam[0]:.2e} / {d_adam[1]:.2e}")
    # both distances are samples of the same chaotic divergence (either may be the larger one): the
    # graphed run has to stay in its range, far below what a wrong step would give
    assert d_graph[0] < 0.1 and d_adam[0] < 0.1, (d_graph, d_adam)
    # the schedule stepped after the first learn(): the update graph in use was captured at the new rate
    lr_now = float(graphed.optimizer.param_groups[0]["lr"])        # after the second scheduler step
    lr_captured = graphed._graphed[32].lr                          # the rate of the second learn() call
    assert lr_captured is not None and lr_now < lr_captured < 1e-3, (lr_now, lr_captured)
    graphed.save_iteration_checkpoint(1)
    other = ParallelTrainer(AlphaZeroNet(num_blocks=2, num_filters=32), device="cpu", checkpoint_dir=str(tmp_path / "g"),
                            verbose=False)
    assert other.load_iteration_checkpoint(1)
    assert not any(g.get("capturable") for g in other.optimizer.param_groups)

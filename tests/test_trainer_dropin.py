"""Training step, replay buffer and on-disk formats of the ParallelTrainer drop-in
against fixtures recorded from the reference (tests/golden/train_tier.npz, made by
tests/golden/make_golden.py train).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from alphazero_amd.config import Config
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.parallel_trainer import (MIRROR_PERM, ExampleArrays, GameDataset, ParallelTrainer,
                                            augment_example, gather_batch)

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "train_tier.npz"))


def train_examples(n=48, seed=5):
    """The generator's synthetic examples (same recipe as make_golden.train_examples)."""
    rs = np.random.RandomState(seed)
    ex = []
    for _ in range(n):
        st = (rs.rand(3, 8, 8) > 0.6).astype(np.float32)
        st[2] = 1.0
        pol = rs.rand(192).astype(np.float32) ** 4
        pol /= pol.sum()
        wl = np.array([1.0, 0.0] if rs.rand() < 0.5 else [0.0, 1.0], np.float32)
        ex.append((st, pol, wl))
    return ex


def structure(obj):
    if isinstance(obj, torch.Tensor):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype)}
    if isinstance(obj, dict):
        return {str(k): structure(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [structure(v) for v in obj]
    return obj


@pytest.fixture
def trained(tmp_path):
    torch.set_num_threads(4)
    torch.manual_seed(123)
    model = AlphaZeroNet(num_blocks=1, num_filters=8)
    tr = ParallelTrainer(model, device="cpu", num_parallel_games=2, num_simulations=4,
                         checkpoint_dir=str(tmp_path), verbose=False)
    tr.add_examples(train_examples())
    torch.manual_seed(7)
    stats = tr.learn(epochs=2, batch_size=16)
    return tr, stats


def test_mirror_augmentation_matches_reference():
    ex = train_examples()
    for i in range(6):
        s, p, w = augment_example(*ex[i])
        assert np.array_equal(s, GOLD["aug_states"][i])
        assert np.array_equal(p, GOLD["aug_policies"][i])
        assert np.array_equal(w, GOLD["aug_wls"][i])
    assert np.array_equal(MIRROR_PERM[MIRROR_PERM], np.arange(192))      # an involution
    # forward moves stay forward, diagonals trade places
    assert MIRROR_PERM[0] == 7 * 3 and MIRROR_PERM[1] == 7 * 3 + 2 and MIRROR_PERM[2] == 7 * 3 + 1


def test_device_gather_equals_dataset_items():
    ex = ExampleArrays.of(train_examples(10))
    ds = GameDataset(ex, augment=True)
    assert len(ds) == 20 and len(GameDataset(ex, augment=False)) == 10
    idx = torch.tensor([0, 13, 9, 10, 19, 4])
    s, p, w = gather_batch(torch.from_numpy(ex.states), torch.from_numpy(ex.policies), torch.from_numpy(ex.wls),
                           idx, torch.from_numpy(MIRROR_PERM))
    for row, i in enumerate(idx.tolist()):
        ds_s, ds_p, ds_w = ds[i]
        assert torch.equal(s[row], ds_s) and torch.equal(p[row], ds_p) and torch.equal(w[row], ds_w)


def test_learn_matches_reference_weights(trained):
    tr, stats = trained
    sd = tr.model.state_dict()
    assert list(sd.keys()) == [str(k) for k in GOLD["learn_keys"]]
    sums = np.array([float(v.double().sum()) for v in sd.values()])
    abs_sums = np.array([float(v.double().abs().sum()) for v in sd.values()])
    # same ops in the same order as the reference: equal up to the CPU's reduction order
    np.testing.assert_allclose(sums, GOLD["learn_sums"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(abs_sums, GOLD["learn_abs_sums"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(sd["input_conv.weight"].numpy(), GOLD["learn_input_conv"], atol=2e-5)
    np.testing.assert_allclose(sd["value_fc2.weight"].numpy(), GOLD["learn_value_fc2"], atol=2e-5)
    assert tr.optimizer.param_groups[0]["lr"] == pytest.approx(float(GOLD["learn_lr_after"]), rel=1e-12)
    assert set(stats) == {"loss", "pi", "wl"} and stats["loss"] == pytest.approx(stats["pi"] + stats["wl"], rel=1e-5)


def test_checkpoint_schema_and_round_trip(trained):
    tr, _ = trained
    tr.save_iteration_checkpoint(3)
    path = os.path.join(tr.checkpoint_dir, "iteration_3.pt")
    ck = torch.load(path, weights_only=True)
    assert structure(ck) == json.loads(GOLD["checkpoint_structure"].item())
    assert tr.get_latest_iteration() == int(GOLD["latest_iteration"]) == 3
    tr.save_iteration_checkpoint(12)
    assert tr.get_latest_iteration() == 12
    # a fresh trainer restores weights, optimiser moments and schedule position
    torch.manual_seed(999)
    other = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=8), device="cpu",
                            checkpoint_dir=tr.checkpoint_dir, verbose=False)
    assert other.load_iteration_checkpoint(3) is True
    assert other.load_iteration_checkpoint(4) is False
    for (k, a), b in zip(tr.model.state_dict().items(), other.model.state_dict().values()):
        assert torch.equal(a, b), k
    assert other.optimizer.param_groups[0]["lr"] == tr.optimizer.param_groups[0]["lr"]
    assert other.scheduler.last_epoch == 1
    # legacy single-file form: same keys without the iteration
    tr.save_checkpoint("legacy.pt")
    legacy = torch.load(os.path.join(tr.checkpoint_dir, "legacy.pt"), weights_only=True)
    assert set(legacy) == {"state_dict", "optimizer", "scheduler", "config"}
    other.load_checkpoint("legacy.pt")
    other.load_checkpoint("missing.pt")     # reports, does not raise


def test_scheduler_horizon_comes_from_the_code_not_the_file(tmp_path):
    """Mirrors the reference's tests/test_trainer.py:28-64."""
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=8), device="cpu", checkpoint_dir=str(tmp_path),
                         verbose=False)
    assert tr.scheduler.T_max == Config.LR_SCHEDULER_T_MAX == 200
    sched = tr.scheduler.state_dict()
    sched["T_max"] = 50
    torch.save({"state_dict": tr.model.state_dict(), "optimizer": tr.optimizer.state_dict(), "scheduler": sched,
                "iteration": 1, "config": {"num_blocks": 1, "num_filters": 8}},
               os.path.join(str(tmp_path), "iteration_1.pt"))
    assert tr.load_iteration_checkpoint(1)
    assert tr.scheduler.T_max == 200


def test_training_data_file_and_replay_buffer(tmp_path):
    ex = train_examples()
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=8), device="cpu", checkpoint_dir=str(tmp_path),
                         verbose=False)
    tr.append_training_data([])                       # nothing to write, no file
    assert not os.path.exists(tr._get_data_file_path())
    assert tr.load_training_data() == 0 and len(tr.examples) == 0
    tr.append_training_data(ex[:10])
    tr.append_training_data(ExampleArrays.of(ex[10:25]))
    d = np.load(tr._get_data_file_path())
    assert sorted(d.files) == [str(k) for k in GOLD["data_keys"]]
    assert {k: [list(d[k].shape), str(d[k].dtype)] for k in d.files} == json.loads(GOLD["data_shapes"].item())
    assert np.array_equal(d["policies"][24], ex[24][1])
    tr.max_examples = 12
    assert tr.load_training_data() == int(GOLD["data_loaded"]) == 12
    assert np.array_equal(tr.examples[0][1], GOLD["data_loaded_first_policy"])
    tr.add_examples(ex[25:30])                        # capped: the oldest five fall out
    assert len(tr.examples) == int(GOLD["buffer_len_after_add"]) == 12
    assert np.array_equal(tr.examples[0][1], GOLD["buffer_first_policy_after_add"])
    assert np.array_equal(tr.examples[-1][1], ex[29][1])
    tr.examples = ex[:3]                              # plain lists are accepted
    assert len(tr.examples) == 3 and isinstance(tr.examples, ExampleArrays)
    s, p, w = tr.examples[1]
    assert s.shape == (3, 8, 8) and p.shape == (192,) and w.shape == (2,)


def test_learn_without_examples_is_a_no_op(tmp_path):
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=8), device="cpu", checkpoint_dir=str(tmp_path),
                         verbose=False)
    before = [p.clone() for p in tr.model.parameters()]
    assert tr.learn(epochs=1, batch_size=8) is None
    assert all(torch.equal(a, b) for a, b in zip(before, tr.model.parameters()))
    assert tr.scheduler.last_epoch == 0


def test_self_play_needs_the_gpu(tmp_path):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=8), device="cpu", num_parallel_games=2,
                         num_simulations=4, checkpoint_dir=str(tmp_path), verbose=False)
    with pytest.raises(Exception, match="GPU|libazamd"):
        tr.execute_parallel_episodes(2)


def test_replay_ring_keeps_the_most_recent_rows_in_order():
    """The ring against the reference's rule (append, keep the last max_examples;
    src/parallel_trainer.py:437-442) on a stream of ragged appends that wraps several times,
    with the cap lowered and raised in between."""
    from alphazero_amd.parallel_trainer import ReplayRing
    rs = np.random.RandomState(3)
    ring = ReplayRing("cpu")
    want = np.zeros((0,), np.float32)
    tag = 0
    limit = 1500
    for step in range(60):
        k = int(rs.choice([0, 1, 7, 300, 1100, 1700]))
        vals = np.arange(tag, tag + k, dtype=np.float32)
        tag += k
        ex = ExampleArrays(np.broadcast_to(vals[:, None, None, None], (k, 3, 8, 8)).copy(),
                           np.broadcast_to(vals[:, None], (k, 192)).copy(), np.broadcast_to(vals[:, None], (k, 2)).copy())
        if step == 25:
            limit = 400
        if step == 40:
            limit = 2500
        ring.append(ex, limit)
        want = np.concatenate([want, vals])[-limit:]
        got = ring.ordered()
        assert len(ring) == len(want) == len(got)
        assert np.array_equal(got.policies[:, 0], want) and np.array_equal(got.states[:, 1, 2, 3], want)
        assert np.array_equal(got.wls[:, 1], want)
        if len(want):
            idx = torch.tensor([0, len(want) - 1, len(want) // 2])
            assert np.array_equal(ring.p[ring.physical(idx), 5].numpy(), want[idx.numpy()])
    assert ring.rows <= 2500


def test_learn_reads_a_wrapped_ring_like_a_flat_buffer(tmp_path):
    """learn() on a ring whose storage has wrapped gives the weights of learn() on the same
    examples held flat: the batches are gathered through the logical -> physical row map."""
    ex = train_examples(40)

    def trainer():
        torch.manual_seed(123)
        tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=8), device="cpu", checkpoint_dir=str(tmp_path),
                             verbose=False)
        tr.max_examples = 1024
        return tr
    wrapped = trainer()
    for i in range(5):                                   # 5 x 281 rows into 1024: overrun, start moves
        wrapped.add_examples(ExampleArrays.of((ex * 8)[i: i + 281]))
    assert wrapped._replay.start != 0 and len(wrapped.examples) == 1024
    flat = trainer()
    flat.examples = wrapped.examples                     # the same 1024 examples, oldest first, from row 0
    assert flat._replay.start == 0
    assert np.array_equal(flat.examples.policies, wrapped.examples.policies)
    for tr in (wrapped, flat):
        torch.manual_seed(7)
        tr.learn(epochs=1, batch_size=64)
    assert all(torch.equal(a, b) for a, b in zip(wrapped.model.parameters(), flat.model.parameters()))


def test_examples_view_is_lazy_and_mutations_reach_the_buffer():
    """`trainer.examples` reads like the reference's live list (src/parallel_trainer.py:127):
    len() copies nothing out of the ring, extend / += append to the ring itself."""
    from alphazero_amd.parallel_trainer import ParallelTrainer, ExampleArrays
    from alphazero_amd.model import AlphaZeroNet
    tr = ParallelTrainer(AlphaZeroNet(num_blocks=1, num_filters=32), device="cpu", num_parallel_games=2,
                         num_simulations=2, verbose=False)
    tr.max_examples = 50
    rs = np.random.RandomState(0)

    def some(n):
        return ExampleArrays(rs.rand(n, 3, 8, 8).astype(np.float32), rs.rand(n, 192).astype(np.float32),
                             rs.rand(n, 2).astype(np.float32))
    a, b, c = some(20), some(10), some(30)
    tr.add_examples(a)
    view = tr.examples
    assert len(view) == 20 and view._np is None and view._t is None          # nothing was copied for len()
    view.extend(b)
    assert len(tr.examples) == 30 and len(tr._replay) == 30
    tr.examples += c                                                          # 60 appended in all, cap 50
    assert len(tr.examples) == 50 and len(tr._replay) == 50
    want = np.concatenate([a.policies, b.policies, c.policies])[-50:]
    assert np.array_equal(tr.examples.policies, want)
    assert np.array_equal(tr.examples[0][1], want[0]) and np.array_equal(tr.examples[-1][1], want[-1])

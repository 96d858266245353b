"""World-size-2 gloo tests (CPU) of the N > 1 path: game sharding, the ragged
sample all-gather and the gradient all-reduce of the data-parallel training step."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alphazero_amd.distributed import shard_games


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alphazero_amd.distributed import all_gather_samples, train_step, losses
        from alphazero_amd.model import AlphaZeroNet
        torch.set_num_threads(1)
        # ragged all-gather: rank r contributes 3 + 2 r samples tagged with its rank; rank 1 also tests empty
        n = 3 + 2 * rank
        g = torch.Generator().manual_seed(rank)
        states = torch.full((n, 3, 8, 8), float(rank))
        policies = torch.rand((n, 192), generator=g)
        wls = torch.tensor([[1.0, 0.0]] * n) if rank == 0 else torch.tensor([[0.0, 1.0]] * n)
        S, P, W = all_gather_samples(states, policies, wls)
        assert S.shape[0] == sum(3 + 2 * r for r in range(world))
        assert torch.all(S[:3] == 0.0) and torch.all(S[3:] == 1.0)
        assert torch.equal(P[3:] if rank == 1 else P[:3], policies)
        assert torch.all(W[:3, 0] == 1.0) and torch.all(W[3:, 1] == 1.0)
        E = all_gather_samples(states[:0], policies[:0], wls[:0]) if rank == 1 else \
            all_gather_samples(states, policies, wls)
        assert E[0].shape[0] == 3                      # rank 1 sent nothing this time
        # the packed form: ragged blocks of 792-byte rows, one all_gather_into_tensor, rank order kept
        from alphazero_amd.distributed import all_gather_packed
        rows = torch.full((4 - 3 * rank, 792), 17 + rank, dtype=torch.uint8)       # 4 rows from rank 0, 1 from rank 1
        rows[:, 0] = torch.arange(len(rows), dtype=torch.uint8)
        G = all_gather_packed(rows)
        assert G.shape == (5, 792) and G.dtype == torch.uint8
        assert G[:4, 1].eq(17).all() and G[4, 1] == 18 and G[:, 0].tolist() == [0, 1, 2, 3, 0]
        G = all_gather_packed(rows[:0] if rank == 0 else rows)                      # rank 0 sends nothing
        assert G.shape == (1, 792) and G[0, 1] == 18
        assert all_gather_packed(rows[:0]).shape == (0, 792)                        # nobody sends anything
        same = all_gather_packed(torch.full((3, 792), rank, dtype=torch.uint8))     # equal counts: no compaction
        assert same.shape == (6, 792) and same[:3].eq(0).all() and same[3:].eq(1).all()
        # data-parallel step == single-process step on the concatenated batch (no BatchNorm
        # difference here: compare gradients computed in eval-mode statistics)
        torch.manual_seed(0)
        model = AlphaZeroNet(num_blocks=1, num_filters=16)
        ref = AlphaZeroNet(num_blocks=1, num_filters=16)
        ref.load_state_dict(model.state_dict())
        gb = torch.Generator().manual_seed(7)
        X = (torch.rand((8, 3, 8, 8), generator=gb) > 0.5).float()
        PI = torch.softmax(torch.rand((8, 192), generator=gb), 1)
        WL = torch.tensor([[1.0, 0.0], [0.0, 1.0]] * 4)
        lo, hi = shard_games(8, rank, world)
        model.eval()                                    # freeze BN statistics for an exact comparison
        loss, _, _ = losses(model, X[lo:hi], PI[lo:hi], WL[lo:hi])
        model.zero_grad()
        loss.backward()
        from alphazero_amd.distributed import allreduce_gradients
        allreduce_gradients(model, (hi - lo) / 8, bucket_bytes=4096)     # several buckets
        ref.eval()
        rl, _, _ = losses(ref, X, PI, WL)
        ref.zero_grad()
        rl.backward()
        for (k, p), q in zip(model.named_parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-5), k
        # ragged slices (6 + 2): the combined gradient is still the gradient of the mean loss
        # over all 8 examples, not the mean of the two slice means
        cut = 6
        mine = slice(0, cut) if rank == 0 else slice(cut, 8)
        loss, _, _ = losses(model, X[mine], PI[mine], WL[mine])
        model.zero_grad()
        loss.backward()
        allreduce_gradients(model, (cut if rank == 0 else 8 - cut) / 8)
        for (k, p), q in zip(model.named_parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, atol=1e-5), k
        # the full step keeps the replicas identical: parameters AND BatchNorm buffers
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        train_step(model, opt, X[mine], PI[mine], WL[mine])          # global batch found by a count exchange
        train_step(model, opt, X[lo:hi], PI[lo:hi], WL[lo:hi], global_batch=8)

        def replicas_equal(m):
            for k, v in m.state_dict().items():
                both = [torch.empty_like(v) for _ in range(world)]
                dist.all_gather(both, v.contiguous())
                assert torch.equal(both[0], both[1]), k
        replicas_equal(model)
        assert float(model.input_bn.running_mean.abs().sum()) > 0    # the statistics did move
        # a rank with an empty slice: the step equals a single-process step on the other rank's data
        torch.manual_seed(3)
        a, b = AlphaZeroNet(num_blocks=1, num_filters=16), AlphaZeroNet(num_blocks=1, num_filters=16)
        b.load_state_dict(a.state_dict())
        oa = torch.optim.Adam(a.parameters(), lr=1e-3, weight_decay=1e-4)
        ob = torch.optim.Adam(b.parameters(), lr=1e-3, weight_decay=1e-4)
        sl = slice(0, 5) if rank == 0 else slice(0, 0)
        train_step(a, oa, X[sl], PI[sl], WL[sl], global_batch=5)
        dist.barrier()
        # single-process reference step: the same function with the collectives switched off
        import alphazero_amd.distributed as azd
        initialised = azd.dist.is_initialized
        azd.dist.is_initialized = lambda: False
        try:
            train_step(b, ob, X[0:5], PI[0:5], WL[0:5])
        finally:
            azd.dist.is_initialized = initialised
        for (k, v), u in zip(a.state_dict().items(), b.state_dict().values()):
            assert torch.allclose(v.float(), u.float(), atol=1e-6), k
        replicas_equal(a)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def _learn_worker(rank, world, port, ret, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alphazero_amd.model import AlphaZeroNet
        from alphazero_amd.parallel_trainer import ParallelTrainer
        torch.set_num_threads(1)
        torch.manual_seed(0)
        model = AlphaZeroNet(num_blocks=1, num_filters=8)
        start = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        tr = ParallelTrainer(model, device="cpu", checkpoint_dir=os.path.join(tmp, f"r{rank}"), verbose=False)
        rs = np.random.RandomState(1)                     # every rank holds the same replay buffer
        n = 8
        states = (rs.rand(n, 3, 8, 8) > 0.5).astype(np.float32)
        pol = rs.rand(n, 192).astype(np.float32)
        pol /= pol.sum(1, keepdims=True)
        wls = np.tile(np.array([[1.0, 0.0], [0.0, 1.0]], np.float32), (n // 2, 1))
        tr.add_examples(list(zip(states, pol, wls)))
        torch.manual_seed(11)                             # and the same shuffle
        # 16 augmented examples in batches of 5, 5, 5, 1: in the last one rank 1's slice is empty
        stats = tr.learn(epochs=1, batch_size=5)
        assert np.isfinite(stats["loss"])
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        assert not torch.equal(flat, start)
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1])              # replicas stay identical
        for k, v in model.state_dict().items():           # ... buffers (BatchNorm statistics) included
            both = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(both, v.contiguous())
            assert torch.equal(both[0], both[1]), k
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_learn_keeps_replicas_identical(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_learn_worker, args=(r, world, port, ret, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: 1, 1: 1}


def test_shard_games():
    assert [shard_games(8192, r, 8) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    spans = [shard_games(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_two_rank_gloo_exchange_steps():
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(ret) == {0: 1, 1: 1}

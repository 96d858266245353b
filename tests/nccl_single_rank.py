"""Runs the exchange steps over the real RCCL backend with a world of one rank (what a one-GPU
box can do): the dtypes, shapes and call forms of alphazero_amd.distributed must be ones RCCL
accepts.  Started as a subprocess by tests/test_gpu_selfplay.py; prints OK on success."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    from alphazero_amd import distributed as azdist
    from alphazero_amd.engine import unpack_samples
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import SelfPlay
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=1, num_filters=128).cuda()
    sp = SelfPlay(model, n_games=16, num_simulations=4, seed=2, pool_nodes_per_game=1 << 10)
    for _ in range(120):
        sp.step(restart_finished=True)
    packed = sp.engine.drain_packed()
    assert packed.shape[0] > 100
    gathered = azdist.all_gather_packed(packed)                    # uint8 rows through all_gather_into_tensor
    assert torch.equal(gathered, packed)
    assert azdist.all_gather_packed(packed[:0]).shape == (0, 792)
    s, p, w = unpack_samples(gathered)
    S, P, W = azdist.all_gather_samples(s, p, w)                   # the float form, list all_gather
    assert torch.equal(S, s) and torch.equal(P, p) and torch.equal(W, w)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    before = [q.detach().clone() for q in model.parameters()]
    # world of one: train_step takes the collective branch only when world > 1, so call its parts
    loss, _, _ = azdist.losses(model.train(), s[:64], p[:64], w[:64])
    opt.zero_grad()
    loss.backward()
    azdist.allreduce_gradients(model, 1.0)                         # flat float32 bucket, SUM
    with torch.no_grad():
        azdist.sync_batchnorm_buffers(model, 1.0)                  # float buffers SUM, int64 counters MAX
    opt.step()
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")    # what bench.py reduces
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    torch.cuda.synchronize()
    # bench.py's exchange block as rank 0 of one runs it on cuda tensors
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("az_bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def barrier():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    ex = bench.exchange_block(argparse.Namespace(games=32, blocks=1), dist, AlphaZeroNet(num_blocks=1, num_filters=128), 0, 1,
                              "cuda", "nccl", barrier)
    assert "error" not in ex, ex
    assert ex["backend"] == "nccl" and ex["unpack_ms"] is not None and ex["train_step_ms"] > 0
    print("OK", flush=True)
finally:
    dist.destroy_process_group()

"""Times az_net_forward (the fused tower + heads) at a given batch size."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet, HipInference  # noqa: E402

FLOP_PER_POS = 778_732_032

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=a.blocks, num_filters=128).eval()
    inf = HipInference(model, max_batch=a.batch)
    g = torch.Generator().manual_seed(1)
    mine = torch.randint(0, 2**62, (a.batch,), generator=g, dtype=torch.int64).cuda()
    theirs = torch.randint(0, 2**62, (a.batch,), generator=g, dtype=torch.int64).cuda() & ~mine
    for _ in range(5):
        inf.forward_bits(mine, theirs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        inf.forward_bits(mine, theirs)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    tf = FLOP_PER_POS * a.batch / (ms * 1e-3) / 1e12 * (a.blocks * 2 + 1 + 0.05) / 41.05
    print(f"batch {a.batch} blocks {a.blocks}: {ms:.3f} ms/forward, {a.batch / ms * 1e3:.0f} pos/s, ~{tf:.1f} TFLOP/s")

"""Forward time of every tower form over a range of batch sizes: where the default form should change.

    python tools/form_sweep.py [--iters 200]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet, HipInference  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--batches", type=int, nargs="*", default=[64, 128, 192, 256, 320, 384, 448, 512, 576, 640, 768, 896, 1024])
a = ap.parse_args()
torch.manual_seed(0)
inf = HipInference(AlphaZeroNet(num_blocks=20, num_filters=128).eval(), max_batch=max(a.batches))
g = torch.Generator().manual_seed(1)
print("batch  " + "  ".join(f"form {f:2d}" for f in (1, 2, 16)) + "   (ms per forward; * = fastest)")
for B in a.batches:
    mine = torch.randint(0, 2**62, (B,), generator=g, dtype=torch.int64).cuda()
    theirs = torch.randint(0, 2**62, (B,), generator=g, dtype=torch.int64).cuda() & ~mine
    row = []
    for form in (1, 2, 16):
        inf.set_tower_form(form)
        for _ in range(10):
            inf.forward_bits(mine, theirs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            inf.forward_bits(mine, theirs)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / a.iters)
    best = min(row)
    print(f"{B:5d}  " + "  ".join(f"{t:6.3f}{'*' if t == best else ' '}" for t in row), flush=True)

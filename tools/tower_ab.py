"""A/B of tower kernel forms at one batch size: interleaved rounds in one process on the same
random-init weights and positions, median and minimum forward time per form.

    python tools/tower_ab.py --batch 1024 --forms 4,16 --rounds 12 --iters 200
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet, HipInference  # noqa: E402

FLOP_PER_POS = 778_732_032

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--forms", default="4,16")
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--warm-seconds", type=float, default=2.0)
    a = ap.parse_args()
    forms = [int(f) for f in a.forms.split(",")]
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=20, num_filters=128).eval()
    inf = HipInference(model, max_batch=a.batch)
    g = torch.Generator().manual_seed(1)
    mine = torch.randint(0, 2**62, (a.batch,), generator=g, dtype=torch.int64).cuda()
    theirs = torch.randint(0, 2**62, (a.batch,), generator=g, dtype=torch.int64).cuda() & ~mine
    import time
    t0 = time.time()
    while time.time() - t0 < a.warm_seconds:      # let the clock settle under load
        for _ in range(50):
            inf.forward_bits(mine, theirs)
        torch.cuda.synchronize()
    times = {f: [] for f in forms}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(a.rounds):
        for f in forms:
            inf.set_tower_form(f)
            for _ in range(10):
                inf.forward_bits(mine, theirs)
            e0.record()
            for _ in range(a.iters):
                inf.forward_bits(mine, theirs)
            e1.record()
            torch.cuda.synchronize()
            times[f].append(e0.elapsed_time(e1) / a.iters)
    for f in forms:
        med, mn = statistics.median(times[f]), min(times[f])
        print(f"form {f}: median {med:.4f} ms, min {mn:.4f} ms per forward at batch {a.batch} "
              f"({FLOP_PER_POS * a.batch / (med * 1e-3) / 1e12:.1f} TFLOP/s incl. heads)", flush=True)

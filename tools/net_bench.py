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
    ap.add_argument("--filters", type=int, default=128, help="other than 128 (or --se-ratio other than 8): the any-width kernel")
    ap.add_argument("--se-ratio", type=int, default=8)
    a = ap.parse_args()
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=a.blocks, num_filters=a.filters, se_ratio=a.se_ratio).eval()
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
    conv = 2 * 64 * 9 * a.filters * a.filters          # FLOP of one 3x3 conv over a position
    flop = conv * (2 * a.blocks + 1) + 2 * 64 * 27 * a.filters + 2 * 64 * a.filters * 32 + 2 * 64 * a.filters * 192 + 2 * 2048 * 128
    if a.filters == 128 and a.se_ratio == 8:
        flop = FLOP_PER_POS * (a.blocks * 2 + 1 + 0.05) / 41.05
    tf = flop * a.batch / (ms * 1e-3) / 1e12
    if os.environ.get("AZ_TOWER_STAMPS"):
        import ctypes as C
        import numpy as np
        from alphazero_amd import _lib
        wg = (a.batch + 3) // 4
        buf = np.zeros((wg, 10), np.uint64)
        L = _lib.lib()
        L.az_net_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        _lib.check(L.az_net_debug_stamps(inf.h, buf.ctypes.data_as(C.c_void_p), wg))
        m = buf.astype(np.float64).mean(0)
        tot = m[0] + m[1] + m[2]
        ticks, m[3] = m[3], a.blocks * 2 + 1
        print(f"stamps (cycles per workgroup, mean): conv loops {m[0]:.0f} ({m[0]/tot*100:.1f}%), conv1 epilogues {m[1]:.0f} "
              f"({m[1]/tot*100:.1f}%), block epilogues {m[2]:.0f} ({m[2]/tot*100:.1f}%); per conv loop {m[0]/m[3]:.0f}, "
              f"per conv1 epilogue {m[1]/((m[3]+1)/2):.0f}, per block epilogue {m[2]/((m[3]-1)/2):.0f}; "
              f"block epilogue parts per block: squeeze {m[4]/((m[3]-1)/2):.0f}, fc1 {m[5]/((m[3]-1)/2):.0f}, "
              f"fc2 {m[6]/((m[3]-1)/2):.0f}, scale+residual+store {m[7]/((m[3]-1)/2):.0f}; "
              f"workgroup duration min/mean/max {buf[:,3].min()/100:.1f}/{ticks/100:.1f}/{buf[:,3].max()/100:.1f} us, "
              f"first start to last end {(int(buf[:,9].max()) - int(buf[:,8].min()))/100:.1f} us, start skew {(int(buf[:,8].max()) - int(buf[:,8].min()))/100:.1f} us; "
              f"stamp counter rate {tot / (ticks / 100e6) / 1e9:.3f} GHz over {ticks / 100:.1f} us")
    print(f"batch {a.batch} blocks {a.blocks} filters {a.filters} se_ratio {a.se_ratio}: {ms:.3f} ms/forward, {a.batch / ms * 1e3:.0f} pos/s, ~{tf:.1f} TFLOP/s")

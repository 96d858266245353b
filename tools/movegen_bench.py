"""Move-generation microbench (BASELINE config 2): n random reachable positions,
az_movegen_encode timed with HIP events over back-to-back launches.

Positions come from an on-device random playout driven through the engine API? No:
they are produced on the host by a vectorised numpy random playout written here
(the CPU oracle is test infrastructure and is not used by this tool)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import _lib  # noqa: E402
from alphazero_amd.game import movegen_encode  # noqa: E402


def random_positions(n, seed):
    """Uniform-random legal self-play, all games advanced in lock-step on the GPU:
    the legal masks come from az_movegen_encode itself, a random legal move is picked
    per game with torch, applied with integer ops; every visited position is kept."""
    dev = torch.device("cuda")
    G = 16384
    g = torch.Generator(device=dev).manual_seed(seed)
    w = torch.full((G,), 0xFFFF, dtype=torch.int64, device=dev)
    b = torch.full((G,), -(1 << 48), dtype=torch.int64, device=dev)   # 0xFFFF << 48 as int64
    t = torch.ones(G, dtype=torch.int8, device=dev)
    out_w, out_b, out_t, have = [], [], [], 0
    FILE_A, FILE_H = 0x0101010101010101, -0x7f7f7f7f7f7f7f80
    while have < n:
        out_w.append(w.clone()); out_b.append(b.clone()); out_t.append(t.clone())
        have += G
        r = movegen_encode(w.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64), t.cpu().numpy(),
                           want_planes=False)
        mask, status = r["mask"], r["status"]
        done = status != 0
        probs = mask.float() + 1e-9
        a = torch.multinomial(probs, 1, generator=g).squeeze(1)
        sq, d = a // 3, a % 3
        to = sq + torch.where(d == 0, 8, torch.where(d == 1, 7, 9))
        white_turn = t > 0
        rf = torch.where(white_turn, sq, 63 - sq)
        rt = torch.where(white_turn, to, 63 - to).clamp(0, 63)
        fb, tb = torch.ones_like(w) << rf, torch.ones_like(w) << rt
        nw = torch.where(white_turn, (w & ~fb) | tb, w & ~tb)
        nb = torch.where(white_turn, b & ~tb, (b & ~fb) | tb)
        w = torch.where(done, torch.full_like(w, 0xFFFF), nw)
        b = torch.where(done, torch.full_like(b, -(1 << 48)), nb)
        t = torch.where(done, torch.ones_like(t), -t)
    return (torch.cat(out_w)[:n].contiguous(), torch.cat(out_b)[:n].contiguous(), torch.cat(out_t)[:n].contiguous())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--mask", action="store_true")
    ap.add_argument("--planes", action="store_true")
    ap.add_argument("--host", action="store_true", help="also time the host-to-host path: numpy arrays in through "
                    "alphazero_amd.game.movegen_encode, legal3 / status (and the mask) back as numpy arrays, PCIe copies included")
    a = ap.parse_args()
    w, b, t = random_positions(min(a.n, 4_000_000), seed=1)
    if a.n > w.numel():                      # tile up to the requested size
        rep = (a.n + w.numel() - 1) // w.numel()
        w, b, t = w.repeat(rep)[:a.n].contiguous(), b.repeat(rep)[:a.n].contiguous(), t.repeat(rep)[:a.n].contiguous()
    n = w.numel()
    dev = w.device
    legal3 = torch.empty((n, 3), dtype=torch.int64, device=dev)
    status = torch.empty(n, dtype=torch.int8, device=dev)
    mask = torch.empty((n, 192), dtype=torch.uint8, device=dev) if a.mask else None
    planes = torch.empty((n, 3, 8, 8), dtype=torch.float32, device=dev) if a.planes else None
    p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run():
        _lib.check(_lib.lib().az_movegen_encode(n, p(w), p(b), p(t), p(legal3), p(mask), p(planes), p(status), st))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    bytes_per = 42 + (192 if a.mask else 0) + (768 if a.planes else 0)
    print(f"n={n} mask={a.mask} planes={a.planes}: {us:.2f} us/launch, {n / us:.1f} Mpos/s, "
          f"{bytes_per} B/pos algorithmic -> {n * bytes_per / us / 1e3:.1f} GB/s "
          f"({n * bytes_per / us / 1e3 / 8000 * 100:.1f}% of 8 TB/s)")
    if a.host:
        import time
        wn, bn, tn = w.cpu().numpy().view(np.uint64), b.cpu().numpy().view(np.uint64), t.cpu().numpy()

        def host_run():
            r = movegen_encode(wn, bn, tn, want_mask=a.mask, want_planes=False)
            out = [r["legal3"].cpu().numpy(), r["status"].cpu().numpy()]
            if a.mask:
                out.append(r["mask"].cpu().numpy())
            return out
        host_run()
        t0 = time.perf_counter()
        reps = max(1, min(20, a.iters))
        for _ in range(reps):
            host_run()
        dt = (time.perf_counter() - t0) / reps
        hb = 42 + (192 if a.mask else 0)
        print(f"host arrays in, host arrays out (pageable memory, copies included): {dt * 1e3:.2f} ms per call, "
              f"{n / dt / 1e6:.1f} Mpos/s, {n * hb / dt / 1e9:.1f} GB/s over PCIe")

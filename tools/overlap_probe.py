"""What is hiding the head-GEMM + tree-step tail of a simulation worth?

One engine of G games runs tower -> head GEMM -> tree step strictly one after the other; while
the two small kernels run (~28 us of ~590 us) most of the chip idles.  This probe runs the same
total number of games as TWO engines of G/2 games, each on its own HIP stream and host
thread, with the tower forced to one position per wave (128 workgroups per half at G = 1024),
so that one half's small kernels can run while the other half's tower occupies its half of the
chip.  Prints simulations per second of both arrangements.

    python tools/overlap_probe.py --games 1024 --sims 800 --plies 6
"""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet, HipInference  # noqa: E402
from alphazero_amd.selfplay import SelfPlay  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=1024)
ap.add_argument("--sims", type=int, default=800)
ap.add_argument("--blocks", type=int, default=20)
ap.add_argument("--plies", type=int, default=6)
ap.add_argument("--parts", type=int, default=2)
a = ap.parse_args()
torch.manual_seed(0)
model = AlphaZeroNet(num_blocks=a.blocks, num_filters=128).eval()


def run(parts):
    per = a.games // parts
    sps, streams = [], []
    for i in range(parts):
        inf = HipInference(model, max_batch=max(16, per))
        if parts > 1:
            inf.set_tower_form(4)
        sps.append(SelfPlay(inf, n_games=per, num_simulations=a.sims, seed=10 + i))
        streams.append(torch.cuda.Stream())

    def play(i, plies):
        with torch.cuda.stream(streams[i]):
            for _ in range(plies):
                sps[i].step(restart_finished=True)
            streams[i].synchronize()

    def all_play(plies):
        ts = [threading.Thread(target=play, args=(i, plies)) for i in range(parts)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    all_play(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    all_play(a.plies)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sims = parts * per * a.sims * a.plies
    print(f"{parts} engine(s) x {per} games: {sims / dt:,.0f} sims/s ({dt / a.plies / a.sims * 1e6:.1f} us per simulation step "
          f"of all {parts * per} games)", flush=True)
    del sps


run(1)
run(a.parts)
run(1)

"""Latency of the single-game search the reference's arena and web front end run
(MCTS.search(game, add_noise=False), fresh root, 200 simulations: src/arena.py:248-265,
src/web.py:447-470) through the drop-in MCTS, for leaf_batch = 1 (the reference's search) and
leaf-parallel settings; and how often the move picked (first child with the most visits)
equals the exact search's on the ten fixture roots of tests/golden/latency_tier.npz.

    python tools/latency_bench.py --sims 200 --leaf-batches 1,4,8,16,32
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alphazero_amd.game import BreakthroughGame, bits_to_board  # noqa: E402
from alphazero_amd.mcts import MCTS  # noqa: E402
from alphazero_amd.model import AlphaZeroNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sims", type=int, default=200)
ap.add_argument("--leaf-batches", default="1,4,8,16,32")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
d = np.load(os.path.join(ROOT, "tests", "golden", "latency_tier.npz"))
games = [BreakthroughGame(board=bits_to_board(int(d["white"][g]), int(d["black"][g])), turn=int(d["turn"][g]))
         for g in range(len(d["white"]))]
torch.manual_seed(0)
model = AlphaZeroNet(num_blocks=20, num_filters=128).cuda().eval()


def best(root):
    acts = list(root.children.keys())
    return acts[int(np.argmax([root.children[x].visit_count for x in acts]))]


exact = None
for k in [int(x) for x in a.leaf_batches.split(",")]:
    m = MCTS(model, num_simulations=a.sims, device="cuda", leaf_batch=k)
    m.search(games[0].clone())                         # warm-up: engine, weights upload
    torch.cuda.synchronize()
    times, picks = [], []
    for g in games:
        for rep in range(a.reps):
            t0 = time.perf_counter()
            root = m.search(g.clone(), add_noise=False)
            mv = best(root)
            times.append(time.perf_counter() - t0)
        picks.append(mv)
    if exact is None:
        exact = picks
    agree = sum(int(x == y) for x, y in zip(picks, exact))
    ms = np.array(times) * 1e3
    print(f"leaf_batch {k:2d}: {a.sims} simulations in {np.median(ms):.1f} ms (median; min {ms.min():.1f}) = "
          f"{np.median(ms) / a.sims * 1e3:.0f} us per simulation; move equals the leaf_batch-1 move on {agree} of "
          f"{len(games)} roots", flush=True)

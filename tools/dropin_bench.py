"""Times the drop-in MCTS.search_batch (host BreakthroughGame objects in, Node handles out)
at the self-play configuration: what a caller of the reference's API sees, host hand-off
included."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from alphazero_amd.game import BreakthroughGame  # noqa: E402
from alphazero_amd.mcts import MCTS  # noqa: E402
from alphazero_amd.model import AlphaZeroNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=1024)
ap.add_argument("--sims", type=int, default=800)
ap.add_argument("--plies", type=int, default=3)
a = ap.parse_args()
torch.manual_seed(0)
np.random.seed(0)
model = AlphaZeroNet(num_blocks=20, num_filters=128).cuda().eval()
mcts = MCTS(model, num_simulations=a.sims, device="cuda")
games = [BreakthroughGame() for _ in range(a.games)]
roots = [None] * a.games
for ply in range(a.plies + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roots = mcts.search_batch(games, roots, add_noise=True)
    t1 = time.perf_counter()
    for i, (g, r) in enumerate(zip(games, roots)):
        probs = mcts.get_action_probs(r, temperature=1.0)
        act = int(np.random.choice(len(probs), p=probs))
        g.step(g.decode_action(act))
        roots[i] = r.children[act] if act in r.children else None
    t2 = time.perf_counter()
    tag = "warm-up" if ply == 0 else f"ply {ply}"
    print(f"{tag}: search_batch {t1 - t0:.3f} s = {a.games * a.sims / (t1 - t0):.0f} sims/s; "
          f"host move loop (probs, choice, step, re-root) {t2 - t1:.3f} s; "
          f"whole ply {a.games * a.sims / (t2 - t0):.0f} sims/s")

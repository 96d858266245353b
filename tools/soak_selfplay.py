"""Plays complete games at the benchmark configuration and reports pool usage, sample
counts and throughput over whole games (not only the opening plies bench.py times)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet  # noqa: E402
from alphazero_amd.selfplay import SelfPlay  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=1024)
ap.add_argument("--sims", type=int, default=800)
ap.add_argument("--blocks", type=int, default=20)
ap.add_argument("--continuous", type=int, default=0, help="play this many plies with finished games restarting instead of one episode batch")
a = ap.parse_args()
torch.manual_seed(0)
sp = SelfPlay(AlphaZeroNet(num_blocks=a.blocks, num_filters=128).eval(), n_games=a.games, num_simulations=a.sims, seed=5)
e = sp.engine
e.reset_all()
e.set_active(np.ones(a.games, np.uint8))
if a.continuous:
    t0 = time.perf_counter()
    last = t0
    for ply in range(1, a.continuous + 1):
        sp.step(restart_finished=True)
        if ply % 16 == 0 and e.samples_pending() > 32768:
            s_, p_, wl_ = e.drain_samples()
            assert bool(torch.all(wl_.sum(1) == 1.0)) and bool(torch.allclose(p_.sum(1), torch.ones(len(p_), device=p_.device), atol=1e-5))
        if ply % 32 == 0:
            st = e.stats()
            now = time.perf_counter()
            print(f"ply {ply}: {32 * a.games * a.sims / (now - last):.0f} sims/s over the last 32 plies, finished {st['games_finished']}, "
                  f"pool peak {int(e.pool_usage().max())}, samples pending {st['samples_pending']}", flush=True)
            last = now
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = e.stats()
    print(f"continuous: {a.continuous} plies, {st['simulations']} simulations in {dt:.1f} s = {st['simulations'] / dt:.0f} sims/s, "
          f"{st['moves'] / dt:.0f} positions/s, {st['games_finished']} games finished")
    sys.exit(0)
t0 = time.perf_counter()
plies = 0
peak = 0
while e.stats()["games_finished"] < a.games:
    e.play_move(a.sims, add_noise=True, restart_finished=False)
    plies += 1
    if plies % 8 == 0:
        peak = max(peak, int(e.pool_usage().max()))
        st = e.stats()
        print(f"ply {plies}: finished {st['games_finished']}, pool peak {peak}, mean depth {e.mean_depth():.2f}", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = e.stats()
s, p, wl = e.drain_samples()
print(f"{a.games} games to the end: {plies} plies, {st['moves']} moves, {len(s)} samples, {st['simulations']} simulations "
      f"in {dt:.1f} s = {st['simulations'] / dt:.0f} sims/s, {st['moves'] / dt:.0f} positions/s; "
      f"terminal leaves {st.get('terminal_leaves', 'n/a')}, pool peak {max(peak, int(e.pool_usage().max()))} of {1 << 18}")
assert len(s) == st["moves"]
assert bool(torch.all(wl.sum(1) == 1.0)) and bool(torch.allclose(p.sum(1), torch.ones(len(p), device=p.device), atol=1e-5))

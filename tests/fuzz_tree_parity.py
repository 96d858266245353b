"""Differential fuzz: the device search engine against the C oracle on random scenarios.

Each round draws its own lookup-table "network" (peaked, flat, all-zero and illegal-only
policy rows; values including exact -1, 0 and 1), start positions from seeded random play,
an exploration constant, a simulation count, noise on or off (numpy Dirichlet draws handed
to both sides), then plays several plies with tree reuse, re-rooting sometimes at a child
that was never visited.  After every search the root statistics of every live game must be
equal bit for bit (children in insertion order, visits, float32 value sums, priors and their
width), as must the positions and the mean search depth.

    python tests/fuzz_tree_parity.py --rounds 200 --seed 1

Needs a GPU.  tests/test_gpu_fuzz.py runs a few rounds of it.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fake_net import mix64, planes_to_bits  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class RandomTable:
    """planes -> (probs f32[n,192], values f32[n]) by table lookup on a hash of the position."""

    def __init__(self, rs, rows=2048):
        sigma = rs.choice([0.3, 2.0, 8.0, 25.0], size=(rows, 1)).astype(np.float32)
        logits = (rs.standard_normal((rows, 192)).astype(np.float32) * sigma).astype(np.float32)
        e = np.exp(logits - logits.max(1, keepdims=True), dtype=np.float32)
        self.probs = (e / e.sum(1, keepdims=True, dtype=np.float32)).astype(np.float32)
        kind = rs.random(rows)
        self.probs[kind < 0.03] = 0.0                       # no mass anywhere: uniform fallback
        only_illegal = (kind >= 0.03) & (kind < 0.05)       # mass on never-legal actions only
        self.probs[only_illegal, :168] = 0.0
        self.values = rs.uniform(-1, 1, rows).astype(np.float32)
        exact = rs.random(rows)
        self.values[exact < 0.05] = -1.0
        self.values[(exact >= 0.05) & (exact < 0.10)] = 1.0
        self.values[(exact >= 0.10) & (exact < 0.15)] = 0.0
        self.rows = rows

    def __call__(self, planes, need=None):
        mine, theirs = planes_to_bits(planes)
        idx = (mix64(mine, theirs) % np.uint64(self.rows)).astype(np.int64)
        return self.probs[idx], self.values[idx]


def same_children(a, b):
    if a is None or b is None:
        return a is None and b is None
    return (np.array_equal(a["actions"], b["actions"]) and np.array_equal(a["N"], b["N"])
            and np.array_equal(a["W"].view(np.uint32), b["W"].view(np.uint32))
            and np.array_equal(a["P"].view(np.uint64), b["P"].view(np.uint64))
            and np.array_equal(a["p_is64"], b["p_is64"]) and a["root_N"] == b["root_N"]
            and np.float32(a["root_W"]).view(np.uint32) == np.float32(b["root_W"]).view(np.uint32))


def one_round(seed, n_games=16, max_plies=10, max_sims=160, verbose=False):
    from gpu_backend import EngineBackend
    rs = np.random.RandomState(seed)
    table = RandomTable(rs)
    c_puct = float(rs.choice([0.5, 1.0, 1.5, 2.1, 3.3, 5.0]))
    sims = int(rs.choice([1, 2, 3, 7, 20, 50, max_sims]))
    noisy = bool(rs.random() < 0.5)
    plies = int(rs.randint(1, max_plies + 1))
    w, b, t, win = orc.random_positions(4 * n_games, seed)
    live = [i for i in range(len(w)) if win[i] == 0][:n_games]
    n = len(live)
    ref = orc.OracleMCTS(n, c_puct=c_puct)
    dev = EngineBackend(n, c_puct=c_puct, pool_nodes_per_game=1 << 18)
    pos = []
    for g, i in enumerate(live):
        p = orc.make_pos(w[i], b[i], t[i])
        pos.append(p)
        ref.set_game(g, p)
        dev.set_game(g, p)
    active = np.array([0 if orc.is_terminal(p) else 1 for p in pos], np.uint8)
    checked = 0
    for ply in range(plies):
        if not active.any():
            break
        ref.set_active(active)
        dev.set_active(active)
        noise = None
        if noisy:
            noise = np.zeros((n, 48), np.float64)
            for g in range(n):
                k = len(orc.legal_actions(pos[g]))
                if active[g] and k:
                    noise[g, :k] = rs.dirichlet([0.35] * k)
        ref.search(table, sims, noise=noise)
        dev.search(table, sims, noise=noise)
        for g in range(n):
            if not active[g]:
                continue
            a, d = ref.root_children(g), dev.root_children(g)
            if not same_children(a, d):
                raise AssertionError(f"seed {seed} ply {ply} game {g}: c_puct {c_puct} sims {sims} noise {noisy}\n"
                                     f"oracle {a}\ndevice {d}")
            checked += 1
        if abs(ref.mean_depth() - dev.mean_depth()) > 1e-12:
            raise AssertionError(f"seed {seed} ply {ply}: mean depth {ref.mean_depth()} != {dev.mean_depth()}")
        for g in range(n):
            if not active[g]:
                continue
            ch = ref.root_children(g)
            if rs.random() < 0.8 and ch["N"].sum() > 0:
                act = int(rs.choice(ch["actions"], p=ch["N"] / ch["N"].sum()))
            else:
                act = int(rs.choice(ch["actions"]))            # possibly a child never visited
            ref.advance(g, act)
            dev.advance(g, act)
            pos[g] = ref.get_game(g)
            q = dev.get_game(g)
            if (pos[g].white, pos[g].black, pos[g].turn, pos[g].winner) != (q.white, q.black, q.turn, q.winner):
                raise AssertionError(f"seed {seed} ply {ply} game {g}: positions differ after action {act}")
            if orc.is_terminal(pos[g]):
                active[g] = 0
    if verbose:
        print(f"seed {seed}: c_puct {c_puct} sims {sims} noise {noisy} plies {plies}: {checked} roots equal", flush=True)
    return checked


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--games", type=int, default=16)
    ap.add_argument("--max-sims", type=int, default=160)
    ap.add_argument("--max-plies", type=int, default=10)
    a = ap.parse_args()
    t0 = time.time()
    total = 0
    for r in range(a.rounds):
        total += one_round(a.seed * 100003 + r, n_games=a.games, max_plies=a.max_plies, max_sims=a.max_sims,
                           verbose=True)
    print(f"{a.rounds} rounds, {total} root comparisons, all equal, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

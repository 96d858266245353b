"""Differential fuzz: the fused search (network inside the library, active games packed into
dense network rows, one tree kernel per simulation) against the step API driven with the same
HIP network, on random scenarios.

Per round: a number of games that lands in every size class of the tower kernel, a random
active mask, a tower form forced for both sides (the forms differ in summation order, so the
comparison holds per form), a simulation count, device-drawn root noise on or off (two engines
with one seed draw the same noise) and a few plies of tree reuse.  After every search the
root statistics of every active game must be equal bit for bit and inactive games untouched.

    python tests/fuzz_fused_parity.py --rounds 60 --seed 1

Needs a GPU.  tests/test_gpu_fuzz.py runs a few rounds of it.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import oracle as orc  # noqa: E402

SIZES = (1, 3, 16, 37, 64, 130, 256, 300, 512, 520)
_cache = {}


def _inference(blocks, max_batch=1024):
    from alphazero_amd.model import AlphaZeroNet, HipInference
    if blocks not in _cache:
        torch.manual_seed(100 + blocks)
        _cache[blocks] = HipInference(AlphaZeroNet(num_blocks=blocks, num_filters=128).eval(), max_batch=max_batch)
    return _cache[blocks]


def _equal(a, b):
    if a is None or b is None:
        return a is None and b is None
    return (np.array_equal(a["actions"], b["actions"]) and np.array_equal(a["N"], b["N"])
            and np.array_equal(a["W"].view(np.uint32), b["W"].view(np.uint32))
            and np.array_equal(a["P"].view(np.uint64), b["P"].view(np.uint64))
            and np.array_equal(a["p_is64"], b["p_is64"]) and a["root_N"] == b["root_N"])


def one_round(seed, sizes=SIZES, max_sims=48, max_plies=3, verbose=False):
    from alphazero_amd.engine import Engine
    rs = np.random.RandomState(seed)
    n = int(rs.choice(sizes))
    sims = int(rs.choice([1, 2, 9, max_sims]))
    plies = int(rs.randint(1, max_plies + 1))
    noisy = bool(rs.random() < 0.5)
    form = int(rs.choice([1, 2, 4, 8, 16]))
    inf = _inference(int(rs.choice([1, 2])))
    inf.set_tower_form(form)
    w, b, t, win = orc.random_positions(3 * n + 8, seed)
    live = np.array([i for i in range(len(w)) if win[i] == 0
                     and orc.legal_actions(orc.make_pos(w[i], b[i], t[i])).size][:n])
    n = len(live)
    eng_seed = int(rs.randint(1 << 30))
    fused = Engine(n, pool_nodes_per_game=1 << 15, seed=eng_seed)
    step = Engine(n, pool_nodes_per_game=1 << 15, seed=eng_seed)
    for e in (fused, step):
        e.set_positions(np.arange(n), w[live], b[live], t[live])
    fused.attach_net(inf)
    start = torch.from_numpy(orc.encode_state(orc.start_position())).cuda()

    def evaluate(planes, need):
        x = planes.clone()
        x[need == 0] = start          # rows nobody asks for hold no planes: any valid state
        probs, value, _, _ = inf.forward_planes(x)
        return probs, value

    if rs.random() < 0.3:
        active = np.ones(n, np.uint8)
    else:
        active = (rs.random(n) < rs.choice([0.15, 0.5, 0.9])).astype(np.uint8)
    checked = 0
    try:
        for ply in range(plies):
            fused.set_active(active)
            step.set_active(active)
            fused.search(sims, add_noise=noisy)       # a no-op (no noise drawn either) with no active game
            if active.any():
                step.search_with(evaluate, sims, add_noise=noisy)
            slots = np.arange(n)
            fa, sa = fused.roots_children(slots), step.roots_children(slots)
            acts = np.full(n, -1, np.int16)
            for g in range(n):
                if not _equal(fa[g], sa[g]):
                    raise AssertionError(f"seed {seed} ply {ply} game {g} of {n} (active {int(active.sum())}): form {form} "
                                         f"sims {sims} noise {noisy}\nfused {fa[g]}\nstep  {sa[g]}")
                if active[g]:
                    assert fa[g] is not None and fa[g]["root_N"] >= sims
                    checked += 1
                    ch = fa[g]
                    acts[g] = int(ch["actions"][np.argmax(ch["N"])]) if rs.random() < 0.7 else int(rs.choice(ch["actions"]))
            if not np.array_equal(fused.pool_usage(), step.pool_usage()):
                raise AssertionError(f"seed {seed} ply {ply}: node pools differ")
            fused.advance_all(acts)
            step.advance_all(acts)
            fw, sw = fused.get_positions(), step.get_positions()
            for x, y in zip(fw, sw):
                assert np.array_equal(x, y)
            active = active & (fw[3] == 0).astype(np.uint8)
            for g in np.nonzero(active)[0]:          # a side without moves ends the game as well
                if not orc.legal_actions(orc.make_pos(fw[0][g], fw[1][g], fw[2][g])).size:
                    active[g] = 0
            if rs.random() < 0.3:                     # wake or park some games between plies
                flip = rs.random(n) < 0.2
                active = np.where(flip & (fw[3] == 0), 1 - active, active).astype(np.uint8)
                for g in np.nonzero(active)[0]:
                    if not orc.legal_actions(orc.make_pos(fw[0][g], fw[1][g], fw[2][g])).size:
                        active[g] = 0
    finally:
        inf.set_tower_form(0)
        fused.close()
        step.close()
    if verbose:
        print(f"seed {seed}: {n} games form {form} sims {sims} noise {noisy} plies {plies}: {checked} roots equal", flush=True)
    return checked


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-sims", type=int, default=48)
    a = ap.parse_args()
    t0 = time.time()
    total = 0
    for r in range(a.rounds):
        total += one_round(a.seed * 100003 + r, max_sims=a.max_sims, verbose=True)
    print(f"{a.rounds} rounds, {total} root comparisons, all equal, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

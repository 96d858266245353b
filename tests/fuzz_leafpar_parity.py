"""Fuzz of the leaf-parallel latency search against its CPU model (tests/leafpar_model.py): random
networks (1-3 blocks, policy layer scaled x1 to x12 so that priors run from flat to sharply peaked),
random mid-game positions of both colours, batches of 2 to 32 descents, 1 to 160 simulations, noisy
and plain roots, several exploration constants; whole trees compared bit for bit.

    python tests/fuzz_leafpar_parity.py --rounds 200 --seed 1

Needs a GPU."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from test_gpu_leafpar import run_case
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    t0 = time.time()
    nodes = 0
    tot = dict(terminal=0, shared=0, evaluated=0)
    for r in range(a.rounds):
        seed = a.seed * 100003 + r
        rs = np.random.RandomState(seed % (1 << 31))
        K = int(rs.choice([2, 3, 4, 7, 8, 16, 31, 32]))
        S = int(rs.choice([1, 2, K, K + 1, 3 * K, int(rs.randint(1, 161))]))
        noise = bool(rs.randint(2))
        blocks = int(rs.choice([1, 2, 3]))
        scale = float(rs.choice([1.0, 4.0, 8.0, 12.0]))
        cp = float(rs.choice([0.7, 1.5, 2.1, 4.0]))
        n, st = run_case(K, S, noise, seed=seed % 1000003, blocks=blocks, scale=scale, n=int(rs.choice([1, 2, 4, 8])), c_puct=cp)
        nodes += n
        for k in tot:
            tot[k] += st[k]
        print(f"seed {seed}: K {K} S {S} noise {noise} blocks {blocks} scale {scale} c_puct {cp}: {n} nodes equal, {st}", flush=True)
    print(f"{a.rounds} rounds, {nodes} tree nodes compared, all equal; descents: {tot}; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

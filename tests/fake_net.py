"""Deterministic table-lookup evaluator used by the bit-exact tree tests.

The "network" is a lookup: a 64-bit mix of the encoded position selects one of
T stored (policy probabilities, value) rows.  The rows were produced once by the
recipe of the reference's MCTS._batch_evaluate (mcts.py:252-256: torch CPU
softmax over float32 logits, value = P(win) - P(loss)) and are stored in
tests/golden/fake_table.npz, so every consumer (reference, oracle, GPU engine)
sees bit-identical float32 inputs on any machine.
"""
import os

import numpy as np

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def planes_to_bits(planes):
    """[n,3,8,8] float32 -> (mine u64[n], theirs u64[n]); bit = row*8+col."""
    p = np.asarray(planes).reshape(len(planes), 3, 64)
    w = (np.uint64(1) << np.arange(64, dtype=np.uint64))
    mine = ((p[:, 0] > 0.5).astype(np.uint64) * w).sum(axis=1, dtype=np.uint64)
    theirs = ((p[:, 1] > 0.5).astype(np.uint64) * w).sum(axis=1, dtype=np.uint64)
    return mine, theirs


def mix64(mine, theirs):
    with np.errstate(over="ignore"):
        x = mine * np.uint64(0x9E3779B97F4A7C15) ^ (theirs + np.uint64(0x7F4A7C15F39CC060)) * np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(31)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(29)
    return x


class TableEvaluator:
    def __init__(self, path=None):
        d = np.load(path or os.path.join(_GOLDEN, "fake_table.npz"))
        self.probs = d["probs"]      # [T,192] float32
        self.values = d["values"]    # [T] float32
        self.T = len(self.values)

    def index(self, planes):
        mine, theirs = planes_to_bits(planes)
        return (mix64(mine, theirs) % np.uint64(self.T)).astype(np.int64)

    def __call__(self, planes, need=None):
        idx = self.index(planes)
        return self.probs[idx], self.values[idx]

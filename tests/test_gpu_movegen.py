"""GPU parity: az_movegen_encode vs the CPU oracle and the reference-generated
fixtures.  Bit-exact (integer / 0-1 float outputs)."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _run(white, black, turn, want_mask=True, want_planes=True):
    from alphazero_amd import _lib
    dev = torch.device("cuda:0")
    n = len(white)
    tw = torch.from_numpy(white.view(np.int64).copy()).to(dev)
    tb = torch.from_numpy(black.view(np.int64).copy()).to(dev)
    tt = torch.from_numpy(turn.copy()).to(dev)
    legal3 = torch.zeros((n, 3), dtype=torch.int64, device=dev)
    mask = torch.zeros((n, 192), dtype=torch.uint8, device=dev) if want_mask else None
    planes = torch.zeros((n, 3, 8, 8), dtype=torch.float32, device=dev) if want_planes else None
    status = torch.zeros(n, dtype=torch.int8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(_lib.lib().az_movegen_encode(n, p(tw), p(tb), p(tt), p(legal3), p(mask), p(planes),
                                            p(status), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return (legal3.cpu().numpy().view(np.uint64),
            None if mask is None else mask.cpu().numpy(),
            None if planes is None else planes.cpu().numpy(),
            status.cpu().numpy())


def test_movegen_matches_reference_fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "game_tier.npz"))
    legal3, mask, planes, status = _run(d["white"], d["black"], d["turn"])
    assert np.array_equal(np.packbits(mask.astype(bool), axis=1), d["mask"])
    assert np.array_equal(np.packbits(planes.reshape(len(planes), -1) > 0.5, axis=1), d["planes"])
    assert set(np.unique(planes)) <= {0.0, 1.0}
    won = d["winner"] != 0
    assert np.array_equal(status[won], d["winner"][won])
    assert np.all(status[(d["k"] == 0) & ~won] == 2)
    assert np.all(status[(d["k"] > 0) & ~won] == 0)
    # legal3 <-> ordered action list: same set, and popcount = k
    for i in range(len(status)):
        acts = set(int(a) for a in d["actions"][i][: d["k"][i]])
        got = {sq * 3 + dd for dd in range(3) for sq in range(64) if (int(legal3[i, dd]) >> sq) & 1}
        assert acts == got


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 100_003])
def test_movegen_ragged_sizes_vs_oracle(n):
    if n == 0:
        from alphazero_amd import _lib
        _lib.check(_lib.lib().az_movegen_encode(0, None, None, None, None, None, None, None, None))
        return
    w, b, t, _ = orc.random_positions(n, seed=n)
    legal3, mask, planes, status = _run(w, b, t)
    o3, om, op, os_ = orc.movegen_batch(w, b, t)
    assert np.array_equal(legal3, o3)
    assert np.array_equal(mask, om)
    assert np.array_equal(planes, op)
    assert np.array_equal(status, os_)


def test_movegen_optional_outputs():
    w, b, t, _ = orc.random_positions(5000, seed=3)
    full = _run(w, b, t)
    only_packed = _run(w, b, t, want_mask=False, want_planes=False)
    assert np.array_equal(full[0], only_packed[0]) and np.array_equal(full[3], only_packed[3])
    only_mask = _run(w, b, t, want_mask=True, want_planes=False)
    assert np.array_equal(full[1], only_mask[1])
    only_planes = _run(w, b, t, want_mask=False, want_planes=True)
    assert np.array_equal(full[2], only_planes[2])


def test_movegen_one_million_positions():
    """BASELINE config 2: 1M random reachable positions, bit-exact masks."""
    n = 1_000_000
    w, b, t, _ = orc.random_positions(n, seed=20260109)
    legal3, mask, _, status = _run(w, b, t, want_planes=False)
    o3, om, _, os_ = orc.movegen_batch(w, b, t, want_planes=False)
    assert np.array_equal(legal3, o3)
    assert np.array_equal(status, os_)
    assert hashlib.sha256(mask.tobytes()).hexdigest() == hashlib.sha256(om.tobytes()).hexdigest()
    # size-independent property: popcount(legal3) == number of set mask bytes
    pc = np.zeros(n, np.int64)
    for dd in range(3):
        pc += np.array([bin(int(x)).count("1") for x in legal3[:2000, dd]] + [0] * (n - 2000))
    assert np.array_equal(pc[:2000], mask[:2000].sum(axis=1))


def test_movegen_matches_the_reference_baseline_engine(golden_dir):
    """Second, independently written oracle: the move sets, is_terminal and check_win of the
    reference's bitboard engine (src/baseline/state.py:90-311), which the reference's own
    tests/test_compatibility.py cross-checks against src/game.py."""
    from test_oracle_golden import check_against_baseline_engine
    d = np.load(os.path.join(golden_dir, "baseline_movegen_tier.npz"))
    legal3, _, _, status = _run(d["white"], d["black"], d["turn"], want_mask=False, want_planes=False)
    check_against_baseline_engine(d, legal3, status)

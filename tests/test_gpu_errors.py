"""GPU: error behaviour of the C ABI — every misuse is reported as a negative status
with a message (AzError in the Python wrappers), never as a fault or a silent no-op."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_call_sequence_is_checked():
    from alphazero_amd._lib import AzError
    from alphazero_amd.engine import Engine
    e = Engine(4, pool_nodes_per_game=256)
    e.reset_all()
    e.set_active(np.ones(4, np.uint8))
    with pytest.raises(AzError, match="without az_engine_select"):
        e.expand_backup(e.probs, e.values)
    with pytest.raises(AzError, match="without az_engine_roots_prepare"):
        e.roots_expand(e.probs)
    with pytest.raises(AzError, match="no network attached"):
        e.search(4)


def test_argument_ranges_are_checked():
    from alphazero_amd import _lib
    from alphazero_amd._lib import AzError
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet, HipInference
    e = Engine(4, pool_nodes_per_game=256)
    with pytest.raises(AzError, match="out of range"):
        e.set_positions([4], [0xFFFF], [0xFFFF << 48], [1])
    with pytest.raises(AzError, match="out of range"):
        e.set_roots([9], [0])
    with pytest.raises(AzError, match="out of range"):
        e.node_children(7)
    inf = HipInference(AlphaZeroNet(num_blocks=1, num_filters=128), max_batch=16)
    with pytest.raises(AzError, match="exceeds max_batch"):
        inf.forward_planes(torch.ones((17, 3, 8, 8), device="cuda"))
    L = _lib.lib()
    with pytest.raises(AzError, match="n_games"):
        Engine(0)
    with pytest.raises(AzError, match="pool_nodes_per_game"):
        Engine(4, pool_nodes_per_game=8)
    # movegen: a mask pointer that is not 16-byte aligned
    w = torch.tensor([0xFFFF], dtype=torch.int64, device="cuda")
    b = torch.tensor([-(1 << 48)], dtype=torch.int64, device="cuda")
    t = torch.ones(1, dtype=torch.int8, device="cuda")
    mask = torch.zeros(256, dtype=torch.uint8, device="cuda")
    rc = L.az_movegen_encode(1, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(t.data_ptr()), None,
                             C.c_void_p(mask.data_ptr() + 4), None, None, _stream())
    assert rc == -1 and b"aligned" in L.az_last_error()
    rc = L.az_movegen_encode(-1, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(t.data_ptr()), None,
                             None, None, None, _stream())
    assert rc == -1
    assert L.az_movegen_encode(0, None, None, None, None, None, None, None, _stream()) == 0   # empty input is fine
    # diagnostics that are switched off answer with an error, not with a read of nothing
    L.az_net_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    buf = np.zeros((4, 10), np.uint64)
    assert L.az_net_debug_stamps(inf.h, buf.ctypes.data_as(C.c_void_p), 4) == -1 and b"stamps are off" in L.az_last_error()
    with pytest.raises(AzError, match="not one of"):
        inf.set_tower_form(3)
    # inputs on the host or of the wrong shape never reach the kernels
    with pytest.raises(AssertionError):
        inf.forward_planes(torch.ones((2, 3, 8, 8)))
    with pytest.raises(AssertionError):
        inf.forward_planes(torch.ones((2, 3, 8, 7), device="cuda"))


def test_capacity_errors_are_reported_once_and_the_engine_stays_usable():
    from alphazero_amd._lib import AzError
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet, HipInference
    torch.manual_seed(1)
    inf = HipInference(AlphaZeroNet(num_blocks=1, num_filters=128).eval(), max_batch=16)
    e = Engine(8, pool_nodes_per_game=128)            # far too small for 64 simulations
    e.attach_net(inf)
    e.reset_all()
    only0 = np.zeros(8, np.uint8)
    only0[0] = 1
    e.set_active(only0)
    with pytest.raises(AzError, match="node pool exhausted"):
        e.search(64)
    # reported once: the flag is cleared behind the report, the other slots of the engine
    # (the drop-in MCTS keeps up to 64 unrelated trees on one engine) are not affected
    e.pool_usage()
    only1 = np.zeros(8, np.uint8)
    only1[1] = 1
    e.set_active(only1)
    e.search(3)                                       # a fresh slot searches fine (22 + 3 x <= 23 nodes fit)
    q = e.root_children(1)
    assert q["root_N"] == 3 and int(q["N"].sum()) == 3
    # the overgrown tree is still a consistent tree: its root has its 22 children and every
    # simulation was backed up
    q0 = e.root_children(0)
    assert len(q0["actions"]) == 22 and q0["root_N"] == 64 and int(q0["N"].sum()) == 64
    # searching it again hits the limit again and says so again
    e.set_active(only0)
    with pytest.raises(AzError, match="node pool exhausted"):
        e.search(8)
    e.pool_usage()


def test_leaf_parallel_search_refuses_a_game_without_a_root():
    """az_engine_search_leaf_parallel descends from existing roots; a slot that has none is
    reported as a call-sequence error (no descent is made for it, no tree is touched), and the
    same call works once the root exists.  Engines come and go: each one carries its own buffers."""
    from alphazero_amd._lib import AzError
    from alphazero_amd.engine import Engine
    from alphazero_amd.model import AlphaZeroNet, HipInference
    torch.manual_seed(3)
    inf = HipInference(AlphaZeroNet(num_blocks=1, num_filters=128).eval(), max_batch=64)
    for _ in range(3):
        e = Engine(4, pool_nodes_per_game=1 << 12)
        e.attach_net(inf)
        e.reset_all()
        with pytest.raises(AzError, match="has no root"):
            e.search_leaf_parallel(np.array([0, 2], np.int32), 16, 8)
        assert e.root_counts().tolist() == [-1, -1, -1, -1]
        e.search(0)                                   # expands the roots
        e.search_leaf_parallel(np.array([0, 2], np.int32), 16, 8)
        q = e.root_children(2)
        assert q["root_N"] == 16 and int(q["N"].sum()) == 16
        assert e.root_children(1)["root_N"] == 0      # a slot that was not named is left alone
        with pytest.raises(AzError, match="appears twice"):
            e.search_leaf_parallel(np.array([1, 1], np.int32), 4, 2)
        with pytest.raises(AzError, match="is not in 1"):
            e.search_leaf_parallel(np.arange(4, dtype=np.int32), 4, 33)
        with pytest.raises(AzError, match="the network takes 64 rows"):
            e.search_leaf_parallel(np.arange(4, dtype=np.int32), 4, 32)
        e.close()


def test_full_sample_queue_drops_whole_games_and_keeps_the_count_consistent():
    """Queue space is handed out in game order by a scan; a game that does not fit is dropped
    whole, the queue length never counts samples that were not written, and draining the
    queue makes room again."""
    from alphazero_amd._lib import AzError
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import SelfPlay
    torch.manual_seed(2)
    nG = 640                                          # queue capacity is max(65536, 160 * n_games) samples
    sp = SelfPlay(AlphaZeroNet(num_blocks=1, num_filters=128).eval(), n_games=nG, num_simulations=1, seed=4,
                  pool_nodes_per_game=1 << 10)
    e = sp.engine
    overflowed = False
    for _ in range(400):
        try:
            sp.step(restart_finished=True)
        except AzError as exc:
            assert "sample queue full" in str(exc)
            overflowed = True
            break
    assert overflowed, "400 plies of 640 restarting games must overflow 102400 queue slots"
    pending = e.samples_pending()
    assert pending <= 160 * nG
    s, p, wl = e.drain_samples()
    assert len(s) == pending
    assert np.allclose(p.sum(1).cpu().numpy(), 1.0, atol=1e-5)           # only complete, written rows are counted
    assert bool((wl.sum(1) == 1.0).all()) and bool((s[:, 2] == 1.0).all())
    assert e.samples_pending() == 0
    sp.step(restart_finished=True)                    # room again: no error



"""GPU: az_net_forward_noalloc (head buffers in the library's static device array, no
allocation) against az_net_forward, through the C ABI.  Runs last: a failure here must not
hide the other tests from `pytest -x`."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _bits(n, seed):
    w, b, t, _ = orc.random_positions(n, seed=seed)
    _, _, planes, _ = orc.movegen_batch(w, b, t, want_mask=False)
    p = planes.reshape(n, 3, 64) > 0.5
    weights = np.uint64(1) << np.arange(64, dtype=np.uint64)
    mine = (p[:, 0].astype(np.uint64) * weights).sum(1, dtype=np.uint64)
    theirs = (p[:, 1].astype(np.uint64) * weights).sum(1, dtype=np.uint64)
    return (torch.from_numpy(mine.view(np.int64)).cuda(), torch.from_numpy(theirs.view(np.int64)).cuda())


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def test_noalloc_forward_equals_forward_on_two_calls_in_a_row():
    from alphazero_amd import _lib
    from alphazero_amd.model import AlphaZeroNet, HipInference
    L = _lib.lib()
    torch.manual_seed(11)
    inf = HipInference(AlphaZeroNet(num_blocks=3, num_filters=128).eval(), max_batch=256)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # outputs are allocated up front, poisoned, and reused by both calls
    outs = {k: [torch.empty((256, 192), device="cuda"), torch.empty(256, device="cuda"),
                torch.empty((256, 192), device="cuda"), torch.empty((256, 2), device="cuda")] for k in ("ref", "got")}
    for B, seed in ((256, 3), (37, 4)):           # the second call is smaller: stale rows stay behind in the array
        mine, theirs = _bits(B, seed)
        for o in outs.values():
            for t in o:
                t.fill_(float("nan"))
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        _lib.check(L.az_net_forward_noalloc(inf.h, B, _ptr(mine), _ptr(theirs), *[_ptr(t) for t in outs["got"]], stream))
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before          # the device's free memory did not move
        _lib.check(L.az_net_forward(inf.h, B, _ptr(mine), _ptr(theirs), *[_ptr(t) for t in outs["ref"]], stream))
        torch.cuda.synchronize()
        for got, ref in zip(outs["got"], outs["ref"]):
            g, r = got[:B].cpu().numpy(), ref[:B].cpu().numpy()
            assert np.isfinite(r).all()
            assert np.array_equal(g.view(np.uint32), r.view(np.uint32))     # same kernels, same rows: bit for bit
            assert torch.isnan(got[B:]).all()                        # nothing written past the batch
    # the limit is enforced with a message, not a fault
    mine, theirs = _bits(257, 5)
    big = [torch.empty((257, 192), device="cuda"), torch.empty(257, device="cuda")]
    rc = L.az_net_forward_noalloc(inf.h, 257, _ptr(mine), _ptr(theirs), _ptr(big[0]), _ptr(big[1]), None, None, stream)
    assert rc == -1 and b"exceeds the static buffers" in L.az_last_error()
    assert L.az_net_forward_noalloc(inf.h, 0, None, None, None, None, None, None, stream) == 0


def test_noalloc_forward_works_on_a_network_created_for_one_position():
    """The static buffers do not depend on max_batch: a caller that only ever uses the
    no-allocation entry point can create the network with max_batch = 1."""
    from alphazero_amd.model import AlphaZeroNet, HipInference
    torch.manual_seed(12)
    model = AlphaZeroNet(num_blocks=2, num_filters=128).eval()
    small, full = HipInference(model, max_batch=1), HipInference(model, max_batch=64)
    mine, theirs = _bits(64, 8)
    probs = torch.empty((64, 192), device="cuda")
    value = torch.empty(64, device="cuda")
    small.forward_bits_into(mine, theirs, probs, value)
    want_p, want_v, _, _ = full.forward_bits(mine, theirs)
    assert torch.equal(probs, want_p) and torch.equal(value, want_v)

"""GPU parity, network tier: bf16 MFMA inference vs the float32 torch forward
(same module as the reference's src/model.py) and vs logits stored by running
the reference itself.  Tolerance (north-star): max-abs 2e-2 on policy
probabilities and value, top-1 move agreement >= 99 %."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TOL = 2e-2


def randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            n = mod.num_features
            mod.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(n, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(n, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(n, generator=g) * 0.1)


def _positions(n, seed):
    w, b, t, win = orc.random_positions(n, seed)
    _, _, planes, _ = orc.movegen_batch(w, b, t, want_mask=False)
    return planes


def _compare(model, planes, min_top1=0.99, ties=None):
    from alphazero_amd.model import HipInference
    model.eval()
    x = torch.from_numpy(planes)
    with torch.no_grad():
        pl, wl = model(x)
        probs_ref = torch.softmax(pl, dim=1).numpy()
        wlp = torch.softmax(wl, dim=1).numpy()
        value_ref = wlp[:, 0] - wlp[:, 1]
    inf = HipInference(model, max_batch=max(len(planes), 16))
    probs, value, pl_d, wl_d = inf.forward_planes(x.cuda(), want_logits=True)
    torch.cuda.synchronize()
    probs, value = probs.cpu().numpy(), value.cpu().numpy()
    err_p = np.abs(probs - probs_ref).max()
    err_v = np.abs(value - value_ref).max()
    err_l = np.abs(pl_d.cpu().numpy() - pl.numpy()).max()
    top1 = (probs.argmax(1) == probs_ref.argmax(1)).mean()
    print(f"max|dprobs|={err_p:.2e} max|dvalue|={err_v:.2e} max|dlogit|={err_l:.2e} top1={top1:.4f}")
    assert np.allclose(probs.sum(1), 1.0, atol=1e-4)
    assert err_p <= TOL and err_v <= TOL
    if ties is not None:
        # small random networks give nearly flat policies: where the best move differs, the float32
        # forward itself must rate the two moves within `ties` of each other
        rows = np.nonzero(probs.argmax(1) != probs_ref.argmax(1))[0]
        gap = probs_ref[rows, probs_ref[rows].argmax(1)] - probs_ref[rows, probs[rows].argmax(1)]
        assert gap.size == 0 or gap.max() <= ties, gap.max()
        assert err_l <= 1e-2
    assert top1 >= min_top1
    return inf


@pytest.mark.parametrize("blocks", [1, 2, 20])
def test_forward_matches_fp32_torch(blocks):
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=blocks, num_filters=128)
    randomize_bn(model, 1)
    _compare(model, _positions(512, seed=blocks))


@pytest.mark.parametrize("B", [1, 3, 4, 5, 17, 1000])
def test_ragged_batch_sizes(B):
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(3)
    model = AlphaZeroNet(num_blocks=2, num_filters=128)
    randomize_bn(model, 4)
    _compare(model, _positions(B, seed=100 + B), min_top1=0.0 if B < 100 else 0.99)


def test_batch_4096_full_network():
    """The forward-only configuration (batch 4096, 20 x 128): tolerance against the float32
    torch forward (run on the GPU for time, cross-checked against the CPU forward on 128 rows spread over the batch),
    and independence of the batch composition (a position's outputs do not depend on its
    neighbours: same bits at batch 4096 and in a batch of 1000)."""
    from alphazero_amd.model import AlphaZeroNet, HipInference
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=20, num_filters=128)
    randomize_bn(model, 1)
    model.eval()
    planes = _positions(4096, seed=77)
    x = torch.from_numpy(planes)
    with torch.no_grad():
        spread = torch.arange(5, 4096, 32)                    # 128 rows from every part of the batch
        pl_cpu, wl_cpu = model(x[spread])
        gm = model.cuda()
        pl, wl = (t.float().cpu() for t in gm(x.cuda()))
        model.cpu()
    assert float((pl[spread] - pl_cpu).abs().max()) < 1e-3 and float((wl[spread] - wl_cpu).abs().max()) < 1e-3
    probs_ref = torch.softmax(pl, 1).numpy()
    wlp = torch.softmax(wl, 1).numpy()
    inf = HipInference(model, max_batch=4096)
    probs, value, _, _ = inf.forward_planes(x.cuda())
    probs, value = probs.cpu().numpy().copy(), value.cpu().numpy().copy()
    assert np.abs(probs - probs_ref).max() <= TOL
    assert np.abs(value - (wlp[:, 0] - wlp[:, 1])).max() <= TOL
    assert (probs.argmax(1) == probs_ref.argmax(1)).mean() >= 0.99
    sub = np.arange(3, 4096, 4)[:1000]
    p2, v2, _, _ = inf.forward_planes(x[sub].contiguous().cuda())
    assert np.array_equal(p2.cpu().numpy().view(np.uint32), probs[sub].view(np.uint32))
    assert np.array_equal(v2.cpu().numpy().view(np.uint32), value[sub].view(np.uint32))
    # the small-batch forms of the tower (channel-split waves: up to 256 and up to 512 positions)
    # compute the same network: equal to the large-batch form up to summation order
    for count in (200, 400):
        p3, v3, _, _ = inf.forward_planes(x[sub[:count]].contiguous().cuda())
        assert np.abs(p3.cpu().numpy() - probs[sub[:count]]).max() < 1e-3
        assert np.abs(v3.cpu().numpy() - value[sub[:count]]).max() < 1e-2


@pytest.mark.parametrize("B", [8, 200, 400, 1024])
def test_forward_is_reproducible_from_the_first_call(B):
    """A cold first forward (weights not yet in any cache, LDS holding whatever ran before)
    must give the same bits as every later one: catches a tap read before its copy landed
    or any use of uninitialised scratch.  One batch size per form of the tower."""
    from alphazero_amd.model import AlphaZeroNet, HipInference
    torch.manual_seed(5)
    model = AlphaZeroNet(num_blocks=20, num_filters=128).eval()
    x = torch.from_numpy(_positions(B, seed=3)).cuda()
    inf = HipInference(model, max_batch=max(16, B))
    p0, v0, _, _ = inf.forward_planes(x)
    p0, v0 = p0.clone(), v0.clone()
    for _ in range(50):
        p, v, _, _ = inf.forward_planes(x)
        assert torch.equal(p, p0) and torch.equal(v, v0)
    other = HipInference(model, max_batch=max(16, B))          # a second instance, cold again
    p, v, _, _ = other.forward_planes(x)
    assert torch.equal(p, p0) and torch.equal(v, v0)


def test_every_tower_form_computes_the_same_network():
    """Forms 1, 2, 4, 8 and 16 of the tower on one batch: each within tolerance of the float32
    torch forward, equal to one another up to summation order, and bit-reproducible."""
    from alphazero_amd.model import AlphaZeroNet, HipInference
    torch.manual_seed(2)
    model = AlphaZeroNet(num_blocks=20, num_filters=128)
    randomize_bn(model, 3)
    model.eval()
    planes = _positions(300, seed=12)
    x = torch.from_numpy(planes)
    with torch.no_grad():
        pl, wl = model(x)
    probs_ref = torch.softmax(pl, 1).numpy()
    wlp = torch.softmax(wl, 1).numpy()
    value_ref = wlp[:, 0] - wlp[:, 1]
    inf = HipInference(model, max_batch=304)
    out = {}
    for form in (1, 2, 4, 8, 16):
        inf.set_tower_form(form)
        p, v, _, _ = inf.forward_planes(x.cuda())
        p, v = p.clone(), v.clone()
        for _ in range(20):
            p2, v2, _, _ = inf.forward_planes(x.cuda())
            assert torch.equal(p2, p) and torch.equal(v2, v), form
        out[form] = (p.cpu().numpy(), v.cpu().numpy())
        assert np.abs(out[form][0] - probs_ref).max() <= TOL and np.abs(out[form][1] - value_ref).max() <= TOL
    for form in (1, 2, 8, 16):
        assert np.abs(out[form][0] - out[4][0]).max() < 1e-3, form
        assert np.abs(out[form][1] - out[4][1]).max() < 1e-2, form
    inf.set_tower_form(0)
    from alphazero_amd._lib import AzError
    with pytest.raises(AzError):
        inf.set_tower_form(3)


def test_matches_reference_logits_fixture(golden_dir):
    """nn_tier.npz holds logits produced by the reference's own AlphaZeroNet
    (torch.manual_seed(0), 20 x 128, BatchNorm statistics randomised with seed 1)."""
    from alphazero_amd.model import AlphaZeroNet, HipInference
    d = np.load(os.path.join(golden_dir, "nn_tier.npz"))
    n = len(d["sel"])
    planes = np.unpackbits(d["planes"])[: n * 192].reshape(n, 3, 8, 8).astype(np.float32)
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=20, num_filters=128)
    randomize_bn(model, 1)
    model.eval()
    with torch.no_grad():
        pl, wl = model(torch.from_numpy(planes))
    # the drop-in module reproduces the reference module (float32, same init stream)
    assert np.allclose(pl.numpy(), d["full_policy_logits"], atol=1e-4)
    assert np.allclose(wl.numpy(), d["full_wl_logits"], atol=1e-4)
    inf = HipInference(model, max_batch=64)
    probs, value, _, _ = inf.forward_planes(torch.from_numpy(planes).cuda())
    ref_p = torch.softmax(torch.from_numpy(d["full_policy_logits"]), 1).numpy()
    ref_wl = torch.softmax(torch.from_numpy(d["full_wl_logits"]), 1).numpy()
    assert np.abs(probs.cpu().numpy() - ref_p).max() <= TOL
    assert np.abs(value.cpu().numpy() - (ref_wl[:, 0] - ref_wl[:, 1])).max() <= TOL


@pytest.mark.parametrize("filters,ratio,blocks", [(32, 8, 2), (64, 4, 3), (96, 8, 1), (128, 4, 2), (160, 16, 1),
                                                  (256, 8, 2), (32, 32, 1), (64, 1, 1)])
def test_other_widths_match_fp32_torch(filters, ratio, blocks):
    """Widths other than 128 filters / 16 hidden units run k_tower_any + k_head_gemm_any
    (src/model.py:107-114 takes any num_filters and se_ratio; the reference's own tests build 2 x 32
    networks): same tolerance against the float32 torch forward, the same bits on every call, and
    rows that do not depend on the rest of the batch."""
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(filters + ratio)
    model = AlphaZeroNet(num_blocks=blocks, num_filters=filters, se_ratio=ratio)
    randomize_bn(model, 2)
    planes = _positions(300, seed=filters)
    inf = _compare(model, planes, min_top1=0.95, ties=2e-3)
    x = torch.from_numpy(planes).cuda()
    p0, v0, _, _ = inf.forward_planes(x)
    p0, v0 = p0.clone(), v0.clone()
    for _ in range(5):
        p, v, _, _ = inf.forward_planes(x)
        assert torch.equal(p, p0) and torch.equal(v, v0)
    sub = torch.arange(7, 300, 3, device="cuda")
    p, v, _, _ = inf.forward_planes(x[sub].contiguous())
    assert torch.equal(p, p0[sub]) and torch.equal(v, v0[sub])
    from alphazero_amd._lib import AzError
    with pytest.raises(AzError):
        inf.set_tower_form(4)          # the forms are variants of the 128-filter kernel
    if filters > 128:
        with pytest.raises(AzError):   # the static head buffers are sized for 128 filters
            inf.forward_bits_into(torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda"),
                                  torch.empty(4, 192, device="cuda"), torch.empty(4, device="cuda"))


def test_small_network_matches_reference_logits_fixture(golden_dir):
    """nn_tier.npz's `small` logits come from the reference's own 2 x 32 AlphaZeroNet
    (torch.manual_seed(0), BatchNorm statistics randomised with seed 1): the network of the
    reference's tests (tests/test_mcts.py:54) on the HIP path."""
    from alphazero_amd.model import AlphaZeroNet, HipInference
    d = np.load(os.path.join(golden_dir, "nn_tier.npz"))
    n = len(d["sel"])
    planes = np.unpackbits(d["planes"])[: n * 192].reshape(n, 3, 8, 8).astype(np.float32)
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=2, num_filters=32)
    randomize_bn(model, 1)
    model.eval()
    with torch.no_grad():
        pl, wl = model(torch.from_numpy(planes))
    assert np.allclose(pl.numpy(), d["small_policy_logits"], atol=1e-4)
    assert np.allclose(wl.numpy(), d["small_wl_logits"], atol=1e-4)
    inf = HipInference(model, max_batch=64)
    probs, value, pl_d, _ = inf.forward_planes(torch.from_numpy(planes).cuda(), want_logits=True)
    ref_p = torch.softmax(torch.from_numpy(d["small_policy_logits"]), 1).numpy()
    ref_wl = torch.softmax(torch.from_numpy(d["small_wl_logits"]), 1).numpy()
    assert np.abs(probs.cpu().numpy() - ref_p).max() <= TOL
    assert np.abs(value.cpu().numpy() - (ref_wl[:, 0] - ref_wl[:, 1])).max() <= TOL
    # no-allocation entry point: same bits
    x = torch.from_numpy(planes).cuda()
    w = (2 ** torch.arange(63, device="cuda", dtype=torch.int64))
    flat_m, flat_t = x[:, 0].reshape(n, 64).long(), x[:, 1].reshape(n, 64).long()
    bits_m = (flat_m[:, :63] * w).sum(1) + torch.where(flat_m[:, 63] > 0, torch.tensor(-2 ** 63, device="cuda"), torch.tensor(0, device="cuda"))
    bits_t = (flat_t[:, :63] * w).sum(1) + torch.where(flat_t[:, 63] > 0, torch.tensor(-2 ** 63, device="cuda"), torch.tensor(0, device="cuda"))
    p2, v2 = torch.empty(n, 192, device="cuda"), torch.empty(n, device="cuda")
    inf.forward_bits_into(bits_m.contiguous(), bits_t.contiguous(), p2, v2)
    assert torch.equal(p2, probs) and torch.equal(v2, value)


def test_rejects_non_state_input():
    from alphazero_amd.model import AlphaZeroNet, HipInference
    from alphazero_amd._lib import AzError
    model = AlphaZeroNet(num_blocks=1, num_filters=128)
    inf = HipInference(model, max_batch=16)
    with pytest.raises(AzError):
        inf.forward_planes(torch.randn(4, 3, 8, 8).cuda())
    with pytest.raises(AzError):
        HipInference(AlphaZeroNet(num_blocks=1, num_filters=48))     # not a multiple of 32

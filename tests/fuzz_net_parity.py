"""Fuzz of the HIP network forward against the float32 torch forward of the same module.

Per round: a block count, fresh random weights with non-trivial BatchNorm statistics (scaled
so that some rounds have sharply peaked policies), a batch size that lands anywhere in or
between the size classes of the tower kernel, one of its forms (in a quarter of the rounds
another width and squeeze-excitation ratio instead: the any-width kernels), positions from
seeded random play.  Checks: max-abs error of policy probabilities and value within the 2e-2 tolerance
(a round beyond it passes only if an ideal bf16 pipeline -- tests/bf16_reference.py: the same roundings
and nothing else, on the CPU -- is beyond it on the same inputs and the HIP error is within a quarter
of that pipeline's: then the operand precision explains it, not the kernel; such rounds are counted and
reported), probabilities summing to one, every row of the batch equal bit for bit to the same position
evaluated alone and at another row of another batch (same form), and the batch reproducible.

    python tests/fuzz_net_parity.py --rounds 60 --seed 1

Needs a GPU.
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

TOL = 2e-2


def _random_model(rs, blocks, filters=128, se_ratio=8):
    from alphazero_amd.model import AlphaZeroNet
    torch.manual_seed(int(rs.randint(1 << 30)))
    model = AlphaZeroNet(num_blocks=blocks, num_filters=filters, se_ratio=se_ratio)
    g = torch.Generator().manual_seed(int(rs.randint(1 << 30)))
    gain = float(rs.choice([0.7, 1.0, 1.3]))
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            n = mod.num_features
            mod.running_mean.copy_(torch.randn(n, generator=g) * 0.2)
            mod.running_var.copy_(torch.rand(n, generator=g) * 1.5 + 0.25)
            mod.weight.data.copy_((torch.rand(n, generator=g) + 0.5) * gain)
            mod.bias.data.copy_(torch.randn(n, generator=g) * 0.2)
    # random initialisation gives nearly flat policies; a scaled policy layer gives peaked ones
    # (the bf16 error of a logit grows with the scale: at x20, peak probability 0.8, the worst
    # probability error over a batch reaches the 2e-2 tolerance, so the sweep stops at x8)
    model.policy_fc.weight.data.mul_(float(rs.choice([1.0, 4.0, 8.0])))
    return model.eval()


def one_round(seed, max_batch=1100, verbose=False):
    from alphazero_amd.model import HipInference
    rs = np.random.RandomState(seed)
    blocks = int(rs.choice([1, 2, 3, 5, 8]))
    B = int(rs.choice([1, 2, 3, 4, 5, 15, 16, 17, 255, 256, 257, 511, 512, 513, int(rs.randint(1, max_batch))]))
    form = int(rs.choice([0, 1, 2, 4, 8, 16]))
    filters, se_ratio = 128, 8
    if rs.randint(4) == 0:   # another width: the any-width kernels (one form; row independence is checked all the same)
        filters = int(rs.choice([32, 64, 96, 128, 160, 192, 224, 256]))
        se_ratio = int(rs.choice([r for r in (1, 2, 4, 8, 16, 32) if (filters, r) != (128, 8)]))
        form = -1
        B = min(B, 300)
    model = _random_model(rs, blocks, filters, se_ratio)
    w, b, t, _ = orc.random_positions(B, seed)
    _, _, planes, _ = orc.movegen_batch(w, b, t, want_mask=False)
    x = torch.from_numpy(planes)
    with torch.no_grad():
        pl, wl = model(x)
        probs_ref = torch.softmax(pl, 1).numpy()
        wlp = torch.softmax(wl, 1).numpy()
    value_ref = wlp[:, 0] - wlp[:, 1]
    inf = HipInference(model, max_batch=max(B, 16))
    try:
        inf.set_tower_form(max(form, 0))
        xd = x.cuda()
        probs, value, _, _ = inf.forward_planes(xd)
        probs, value = probs.clone(), value.clone()
        p2, v2, _, _ = inf.forward_planes(xd)
        assert torch.equal(p2, probs) and torch.equal(v2, value), f"seed {seed}: not reproducible"
        pn, vn = probs.cpu().numpy(), value.cpu().numpy()
        err_p, err_v = float(np.abs(pn - probs_ref).max()), float(np.abs(vn - value_ref).max())
        assert np.allclose(pn.sum(1), 1.0, atol=1e-4), f"seed {seed}: probabilities do not sum to one"
        beyond = False
        if err_p > TOL or err_v > TOL:
            # what the choice of bf16 operands alone does on this round
            from bf16_reference import emulate
            pl_e, wl_e = emulate(model, x)
            pe = torch.softmax(pl_e, 1).numpy()
            we = torch.softmax(wl_e, 1).numpy()
            ideal_p = float(np.abs(pe - probs_ref).max())
            ideal_v = float(np.abs((we[:, 0] - we[:, 1]) - value_ref).max())
            beyond = err_p <= max(TOL, 1.25 * ideal_p) and err_v <= max(TOL, 1.25 * ideal_v)
            assert beyond, (f"seed {seed}: blocks {blocks} x {filters} / {se_ratio} B {B} form {form}: {err_p:.3e} {err_v:.3e} "
                            f"(ideal bf16 pipeline: {ideal_p:.3e} {ideal_v:.3e})")
            print(f"seed {seed}: beyond {TOL:g} with the ideal bf16 pipeline beyond it too: HIP {err_p:.3e} / {err_v:.3e}, "
                  f"ideal {ideal_p:.3e} / {ideal_v:.3e}, peak probability {probs_ref.max():.3f}", flush=True)
        if form:   # a position's result must not depend on its row or on its neighbours
            for i in {0, B - 1, int(rs.randint(B))}:
                p1, v1, _, _ = inf.forward_planes(xd[i:i + 1].contiguous())
                assert torch.equal(p1[0], probs[i]) and torch.equal(v1[0], value[i]), \
                    f"seed {seed}: row {i} of {B} differs from the position alone (form {form})"
            perm = torch.from_numpy(rs.permutation(B)).cuda()
            pp, vp, _, _ = inf.forward_planes(xd[perm].contiguous())
            assert torch.equal(pp, probs[perm]) and torch.equal(vp, value[perm]), \
                f"seed {seed}: results move with the rows (form {form}, B {B})"
    finally:
        inf.close()
    top1 = float((pn.argmax(1) == probs_ref.argmax(1)).mean())
    if verbose:
        print(f"seed {seed}: blocks {blocks} filters {filters} se_ratio {se_ratio} B {B} form {form}: max|dprobs| {err_p:.2e} max|dvalue| {err_v:.2e} "
              f"top1 {top1:.4f} peak prob {probs_ref.max():.2f}", flush=True)
    return err_p, err_v, top1, B, beyond


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    t0 = time.time()
    worst_p = worst_v = 0.0
    agree = total = n_beyond = 0
    for r in range(a.rounds):
        ep, ev, top1, B, beyond = one_round(a.seed * 100003 + r, verbose=True)
        n_beyond += bool(beyond)
        worst_p, worst_v = max(worst_p, ep), max(worst_v, ev)
        agree += top1 * B
        total += B
    print(f"{a.rounds} rounds, {total} positions: worst max|dprobs| {worst_p:.2e}, worst max|dvalue| {worst_v:.2e}, "
          f"top-1 agreement {agree / total:.4f}, {n_beyond} round(s) beyond {TOL:g} that an ideal bf16 pipeline is beyond as well, "
          f"{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()

"""What an ideal bf16 pipeline gives on a fuzz_net_parity round: the torch network evaluated on the
CPU with the roundings the HIP kernels make and nothing else -- BatchNorm folded into the conv
weights, conv and FC weights rounded to bf16, the input of every convolution and of the two head
GEMMs rounded to bf16, float32 accumulation, float32 squeeze-excitation on the unrounded conv output,
the residual taken from the rounded activations -- against the float32 forward.  It separates
the error that the choice of bf16 operands implies from anything a kernel might add: if a fuzz
round is out of tolerance here as well, the precision is the cause.

    python tools/bf16_emulation.py --seed 4102923703          # a seed printed by tests/fuzz_net_parity.py

Needs no GPU.  With --hip (on a GPU box) the HIP forward of the same round is compared with both.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


from bf16_reference import emulate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True, help="the round's seed as fuzz_net_parity prints it")
    ap.add_argument("--hip", action="store_true", help="also run the HIP forward (needs a GPU)")
    a = ap.parse_args()
    from oracle import oracle as orc
    import fuzz_net_parity as fz
    # the fuzzer's draws, in its order
    rs = np.random.RandomState(a.seed)
    blocks = int(rs.choice([1, 2, 3, 5, 8]))
    B = int(rs.choice([1, 2, 3, 4, 5, 15, 16, 17, 255, 256, 257, 511, 512, 513, int(rs.randint(1, 1100))]))
    form = int(rs.choice([0, 1, 2, 4, 8, 16]))
    filters, se_ratio = 128, 8
    if rs.randint(4) == 0:
        filters = int(rs.choice([32, 64, 96, 128, 160, 192, 224, 256]))
        se_ratio = int(rs.choice([r for r in (1, 2, 4, 8, 16, 32) if (filters, r) != (128, 8)]))
        form = -1
        B = min(B, 300)
    model = fz._random_model(rs, blocks, filters, se_ratio)
    w, b, t, _ = orc.random_positions(B, a.seed)
    _, _, planes, _ = orc.movegen_batch(w, b, t, want_mask=False)
    x = torch.from_numpy(planes)
    with torch.no_grad():
        pl, wl = model(x)
    pl_e, wl_e = emulate(model, x)

    def probs_value(pl, wl):
        p = torch.softmax(pl, 1).numpy()
        q = torch.softmax(wl, 1).numpy()
        return p, q[:, 0] - q[:, 1]

    p_ref, v_ref = probs_value(pl, wl)
    p_emu, v_emu = probs_value(pl_e, wl_e)
    print(f"seed {a.seed}: blocks {blocks} filters {filters} se_ratio {se_ratio} B {B} form {form}; peak probability {p_ref.max():.3f}, "
          f"logits span [{float(pl.min()):.1f}, {float(pl.max()):.1f}]")
    i = int(np.abs(p_emu - p_ref).max(1).argmax())
    print(f"ideal bf16 pipeline vs float32: max|dprobs| {np.abs(p_emu - p_ref).max():.3e} (position {i}), max|dvalue| {np.abs(v_emu - v_ref).max():.3e}, "
          f"max|dlogit| {float((pl_e - pl).abs().max()):.3e}, top-1 agreement {(p_emu.argmax(1) == p_ref.argmax(1)).mean():.4f}")
    if a.hip:
        from alphazero_amd.model import HipInference
        for f in ([0, 1, 2, 4, 8, 16] if form >= 0 else [0]):
            inf = HipInference(model, max_batch=max(B, 16))
            inf.set_tower_form(f)
            probs, value, lg, _ = inf.forward_planes(x.cuda(), want_logits=True)
            pn, vn, ln = probs.cpu().numpy(), value.cpu().numpy(), lg.cpu()
            j = int(np.abs(pn - p_ref).max(1).argmax())
            print(f"HIP form {f:2d} vs float32: max|dprobs| {np.abs(pn - p_ref).max():.3e} (position {j}), max|dvalue| {np.abs(vn - v_ref).max():.3e}, "
                  f"max|dlogit| {float((ln - pl).abs().max()):.3e}; vs the ideal bf16 pipeline: max|dprobs| {np.abs(pn - p_emu).max():.3e}, "
                  f"max|dlogit| {float((ln - pl_e).abs().max()):.3e}")
            inf.close()


if __name__ == "__main__":
    main()

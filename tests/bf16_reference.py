"""The torch network evaluated with the roundings the HIP kernels make and nothing else: BatchNorm
folded into the conv weights, conv and FC weights rounded to bf16, the input of every convolution
and of the two head GEMMs rounded to bf16, float32 accumulation, float32 squeeze-excitation on the
unrounded conv output, the residual taken from the rounded activations.  What an ideal bf16
pipeline computes: the error it shows against the float32 forward is the error the choice of
bf16 operands implies, whatever the kernel.  Test infrastructure (CPU, no GPU needed)."""
import torch
import torch.nn.functional as F


def bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def fold(conv_w, bn):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return bf16(conv_w * s.view(-1, 1, 1, 1)), bn.bias - bn.running_mean * s


def emulate(model, x):
    """The network with bf16 operands and float32 accumulation (see the module docstring)."""
    with torch.no_grad():
        w, b = fold(model.input_conv.weight, model.input_bn)
        act = bf16(F.relu(F.conv2d(bf16(x), w, b, padding=1)))
        for blk in model.residual_tower:
            w1, b1 = fold(blk.conv1.weight, blk.bn1)
            w2, b2 = fold(blk.conv2.weight, blk.bn2)
            mid = bf16(F.relu(F.conv2d(act, w1, b1, padding=1)))
            y = F.conv2d(mid, w2, b2, padding=1)
            n, c = y.shape[0], y.shape[1]
            e = blk.se.fc2(F.relu(blk.se.fc1(y.mean((2, 3)))))
            sw, sb = e.split(c, dim=1)
            act = bf16(F.relu(y * torch.sigmoid(sw).view(n, c, 1, 1) + sb.view(n, c, 1, 1) + act))
        wp, bp = fold(model.policy_conv.weight, model.policy_bn)
        p = bf16(F.relu(F.conv2d(act, wp, bp, padding=1)))
        logits = F.linear(p.reshape(p.size(0), -1), bf16(model.policy_fc.weight), model.policy_fc.bias)
        wv, bv = fold(model.value_conv.weight, model.value_bn)
        v = bf16(F.relu(F.conv2d(act, wv, bv)))
        h = F.relu(F.linear(v.reshape(v.size(0), -1), bf16(model.value_fc1.weight), model.value_fc1.bias))
        wl = model.value_fc2(h)
    return logits, wl

"""Per-workgroup duration of k_tower16 from its cycle stamps, grouped by XCD (workgroup id mod 8) and by launch order.

    AZ_TOWER_STAMPS=1 AZ_TOWER_FORM=16 python tools/wg_spread.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import _lib  # noqa: E402
from alphazero_amd.model import AlphaZeroNet, HipInference  # noqa: E402

B = 1024
torch.manual_seed(0)
inf = HipInference(AlphaZeroNet(num_blocks=20, num_filters=128).eval(), max_batch=B)
g = torch.Generator().manual_seed(1)
mine = torch.randint(0, 2**62, (B,), generator=g, dtype=torch.int64).cuda()
theirs = torch.randint(0, 2**62, (B,), generator=g, dtype=torch.int64).cuda() & ~mine
L = _lib.lib()
L.az_net_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
wg = B // 4
acc = np.zeros((wg, 4))
N = 200
for it in range(N + 20):
    inf.forward_bits(mine, theirs)
    if it < 20:
        continue
    buf = np.zeros((wg, 10), np.uint64)
    _lib.check(L.az_net_debug_stamps(inf.h, buf.ctypes.data_as(C.c_void_p), wg))
    t0 = int(buf[:, 8].min())
    acc[:, 0] += (buf[:, 8].astype(np.int64) - t0) / 100.0          # start, us after the first start
    acc[:, 1] += (buf[:, 9].astype(np.int64) - t0) / 100.0          # end
    acc[:, 2] += buf[:, 3].astype(np.float64) / 100.0               # duration
    acc[:, 3] += buf[:, 0].astype(np.float64) + buf[:, 1].astype(np.float64) + buf[:, 2].astype(np.float64)   # shader-clock cycles over the conv sequence
acc /= N
print(f"mean over {N} launches: duration min {acc[:, 2].min():.1f} mean {acc[:, 2].mean():.1f} max {acc[:, 2].max():.1f} us; last end {acc[:, 1].max():.1f} us")
for x in range(8):
    d = acc[x::8, 2]
    cyc = acc[x::8, 3].mean()
    print(f"workgroups with id % 8 == {x}: duration mean {d.mean():.1f} min {d.min():.1f} max {d.max():.1f} us, start {acc[x::8, 0].mean():.2f} us, "
          f"{cyc / 1e3:.1f} k cycles = {cyc / d.mean() / 1e3:.3f} GHz")
order = np.argsort(acc[:, 2])
print("slowest workgroups:", [(int(i), round(float(acc[i, 2]), 1)) for i in order[-8:]])
print("fastest workgroups:", [(int(i), round(float(acc[i, 2]), 1)) for i in order[:8]])
q = np.array_split(np.arange(wg), 8)
print("by id range:", [round(float(acc[idx, 2].mean()), 1) for idx in q])

"""How much of the training step's time is launch overhead?  Times the reference's step
(forward, soft-target losses, backward, clip, Adam) on the 20x128 network eagerly and replayed
from one captured HIP graph (static input buffers, capturable Adam), at the reference's batch
of 1024 (--batch 1024) or at a rank's slice of it (--batch 128: 1024 over 8 data-parallel ranks).

    python tools/train_graph_probe.py --batch 128
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
B = ap.parse_args().batch
torch.manual_seed(0)
dev = torch.device("cuda")
x = (torch.rand((B, 3, 8, 8), device=dev) > 0.7).float()
pi = torch.softmax(torch.rand((B, 192), device=dev), 1)
wl = torch.tensor([[1.0, 0.0]], device=dev).repeat(B, 1)


def make(capturable):
    torch.manual_seed(0)
    m = AlphaZeroNet(num_blocks=20, num_filters=128).to(dev).train()
    o = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4, capturable=capturable)
    return m, o


def step(m, o, autocast):
    o.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        p, w = m(x)
    loss = -torch.sum(pi * F.log_softmax(p.float(), dim=1)) / B + F.cross_entropy(w.float(), wl)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    o.step()
    return loss


for autocast in (False, True):
    torch.backends.cudnn.benchmark = autocast
    m, o = make(False)
    for _ in range(5):
        step(m, o, autocast)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        step(m, o, autocast)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 30
    m, o = make(True)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            step(m, o, autocast)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step(m, o, autocast)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 30
    print(f"batch {B} autocast={autocast}: eager {eager * 1e3:.1f} ms/step ({1 / eager:.1f} steps/s), graph replay {graph * 1e3:.1f} ms/step "
          f"({1 / graph:.1f} steps/s), loss {float(loss):.3f}", flush=True)

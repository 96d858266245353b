"""Times ParallelTrainer.learn (the reference's training step, float32 PyTorch-ROCm) on
synthetic examples resident on the GPU: optimisation steps per second and examples per second."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from alphazero_amd.model import AlphaZeroNet  # noqa: E402
from alphazero_amd.parallel_trainer import ExampleArrays, ParallelTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--examples", type=int, default=32768)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--blocks", type=int, default=20)
ap.add_argument("--bf16", action="store_true", help="autocast the forward to bfloat16")
ap.add_argument("--find", action="store_true", help="let MIOpen time its solvers per layer shape (cudnn.benchmark)")
ap.add_argument("--channels-last", action="store_true", help="keep weights and activations NHWC")
ap.add_argument("--graph", action="store_true", help="replay full batches from captured HIP graphs (ParallelTrainer(graph_step=True))")
a = ap.parse_args()
torch.backends.cudnn.benchmark = a.find
rs = np.random.RandomState(0)
states = (rs.rand(a.examples, 3, 8, 8) > 0.7).astype(np.float32)
pol = rs.rand(a.examples, 192).astype(np.float32)
pol /= pol.sum(1, keepdims=True)
wls = np.tile(np.array([[1.0, 0.0], [0.0, 1.0]], np.float32), (a.examples // 2, 1))
torch.manual_seed(0)
model = AlphaZeroNet(num_blocks=a.blocks, num_filters=128).cuda()
if a.channels_last:
    model = model.to(memory_format=torch.channels_last)
tr = ParallelTrainer(model, device="cuda", checkpoint_dir=tempfile.mkdtemp(), verbose=False,
                     autocast_dtype=torch.bfloat16 if a.bf16 else None, graph_step=a.graph)
tr.add_examples(ExampleArrays(states, pol, wls))
tr.learn(epochs=1, batch_size=a.batch)            # warm-up (MIOpen kernel selection)
torch.cuda.synchronize()
t0 = time.perf_counter()
stats = tr.learn(epochs=1, batch_size=a.batch)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps = -(-2 * a.examples // a.batch)
print(f"learn(){' graph' if a.graph else ''}{' bf16' if a.bf16 else ''}: {2 * a.examples} augmented examples, batch {a.batch}, {a.blocks}x128: {dt:.2f} s, "
      f"{steps / dt:.1f} steps/s, {2 * a.examples / dt:.0f} examples/s, loss {stats['loss']:.3f}")

"""Times ParallelTrainer.execute_parallel_episodes(1024, continuous=True): 1024 slots always busy,
each call returns when 1024 more games have finished (games in progress carry over)."""
import sys, time, tempfile, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd.model import AlphaZeroNet
from alphazero_amd.parallel_trainer import ParallelTrainer
torch.manual_seed(0)
tr = ParallelTrainer(AlphaZeroNet().cuda(), device="cuda", num_parallel_games=1024, num_simulations=800,
                     checkpoint_dir=tempfile.mkdtemp(), seed=1, verbose=False)
for call in range(3):
    t0 = time.perf_counter()
    ex = tr.execute_parallel_episodes(1024, continuous=True)
    dt = time.perf_counter() - t0
    print(f"continuous call {call}: {len(ex)} examples in {dt:.1f} s = {len(ex)/dt:.0f} examples/s", flush=True)

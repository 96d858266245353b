"""Self-play throughput benchmark (BASELINE.json: MCTS simulations/sec and
self-play positions/sec).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE config 4, per GPU): 1024 concurrent Breakthrough games,
800 simulations per move, Dirichlet noise on, SE-ResNet 20x128 with random-init
weights, tree reuse, finished games restart so the batch stays full.  One step
= one ply of every game = 1024 x 800 simulations.  For N > 1 the games are
sharded (one process per GPU, rank-offset seeds, no data-path collective).

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (the fused
SE-ResNet tower) against the dense bf16 MFMA peak using HIP-event timings taken
inside the timed region; `cpu_baseline` times the CPU oracle (C tree + float32
torch forward) on a bounded sample of the same workload on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POSITION = 778_732_032          # whole network, 2*MAC
TOWER_FLOP_PER_POSITION = 775_061_504    # stem + 40 tower convs + SE + policy conv + value conv
PEAK_BF16_DENSE = 2.5e15


def cpu_baseline(games, sims_budget_s, blocks, cores):
    """Reference-equivalent CPU path: oracle tree (oracle/az_oracle.c) + float32
    torch forward, one search_batch from the start position with noise."""
    from oracle import oracle as orc
    from alphazero_amd.model import AlphaZeroNet
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=blocks, num_filters=128).eval()
    m = orc.OracleMCTS(games, c_puct=1.5)
    for g in range(games):
        m.set_game(g, orc.start_position())

    def evaluate(planes, need):
        with torch.no_grad():
            pl, wl = model(torch.from_numpy(planes))
            probs = torch.softmax(pl, dim=1).numpy()
            wlp = torch.softmax(wl, dim=1).numpy()
        return probs, wlp[:, 0] - wlp[:, 1]
    rs = np.random.RandomState(0)
    m.roots_prepare()
    probs, _ = evaluate(m.planes, m.need)
    m.roots_expand(probs)
    m.add_noise(rs.dirichlet([0.35] * 48, size=games), 0.25)
    t0 = time.perf_counter()
    sims = 0
    while True:
        m.select()
        probs, values = evaluate(m.planes, m.need)
        m.expand_backup(probs, values)
        sims += games
        if time.perf_counter() - t0 > sims_budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=sims / dt, unit="sims/s", cores=cores, kind="port",
                sample=f"{games} games x {sims // games} simulations of one search_batch from the start "
                       f"position (noise on), C oracle tree + fp32 torch forward {blocks}x128, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=1024, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800, help="simulations per move")
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the self-play engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import SelfPlay

    torch.manual_seed(0)                       # same random-init weights on every rank
    model = AlphaZeroNet(num_blocks=args.blocks, num_filters=128).eval()
    sp = SelfPlay(model, n_games=args.games, num_simulations=args.sims, seed=1234 + rank)
    eng, inf = sp.engine, sp.inference

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sp.step(restart_finished=True)
    barrier()
    st0 = eng.stats()
    inf.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sp.step(restart_finished=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tower_ms, tower_samples = inf.tower_time()
    inf.set_timing(False)
    st1 = eng.stats()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([st1[k] - st0[k] for k in ("simulations", "moves", "nn_evals")],
                           dtype=torch.float64, device="cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        sims, moves, evals = (float(x) for x in cnt.tolist())
        dist.barrier()
    else:
        sims, moves, evals = (float(st1[k] - st0[k]) for k in ("simulations", "moves", "nn_evals"))

    if rank == 0:
        achieved = TOWER_FLOP_PER_POSITION * args.games / (tower_ms * 1e-3) if tower_ms > 0 else 0.0
        out = {
            "metric": "selfplay_mcts_simulations_per_sec",
            "value": sims / dt,
            "unit": "sims/s",
            "positions_per_sec": moves / dt,
            "nn_evals_per_sec": evals / dt,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic (random-init weights, games from the start position)",
            "config": {"workload": f"full self-play: {args.games} concurrent games/GPU x {args.sims} sims/move, "
                                   f"Dirichlet noise, SE-ResNet {args.blocks}x128, tree reuse, restart on finish",
                       "games_per_gpu": args.games, "sims_per_move": args.sims,
                       "parallelism": f"games sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_BF16_DENSE / 1e12,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_DENSE, "traffic": None,
                         "kernel": "k_tower", "launch_ms": tower_ms, "launch_samples": tower_samples,
                         "flop_per_launch": TOWER_FLOP_PER_POSITION * args.games,
                         "mean_search_depth": eng.mean_depth()},
        }
        if world == 1 and args.cpu_seconds > 0:
            cores = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(128, args.cpu_seconds, args.blocks, cores)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

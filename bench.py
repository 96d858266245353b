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

FLOP_PER_POSITION = 778_732_032          # whole 20x128 network, 2*MAC


def tower_flop_per_position(blocks):
    """FLOPs (2*MAC) of what the fused tower kernel computes for one position:
    stem conv + blocks x (two 3x3 convs + SE) + policy conv + value conv.
    20 blocks: 775,061,504."""
    conv = 2 * 64 * 128 * 128 * 9
    return 2 * 64 * 128 * 27 + blocks * (2 * conv + 12_288) + conv + 2 * 64 * 32 * 128


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
    # AZ_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with one GPU: every
    # rank then shares cuda:0 and the collectives run on CPU tensors.
    backend = os.environ.get("AZ_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    torch.cuda.set_device(0 if rehearsal else local_rank)
    coll_dev = "cpu" if rehearsal else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend)
        else:
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
    for i in range(args.steps):
        sp.step(restart_finished=True)
        # long runs: hand the finished games' samples over before the device queue fills
        # (on the device, as a trainer would take them; part of the timed work)
        if (i + 1) % 16 == 0 and eng.samples_pending() > 32768:
            eng.drain_samples()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tower_ms, tower_samples = inf.tower_time()
    inf.set_timing(False)
    st1 = eng.stats()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([st1[k] - st0[k] for k in ("simulations", "moves", "nn_evals")],
                           dtype=torch.float64, device=coll_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        sims, moves, evals = (float(x) for x in cnt.tolist())
        dist.barrier()
    else:
        sims, moves, evals = (float(st1[k] - st0[k]) for k in ("simulations", "moves", "nn_evals"))

    exchange = None
    if dist is not None:
        # The two exchange steps of an iteration, timed outside the self-play region on a
        # synthetic iteration's worth of samples (games x 64 plies per rank):
        # RCCL all-gather into every rank's replay buffer, then one data-parallel
        # training step at global batch 1024 with an RCCL gradient all-reduce.
        try:
            from alphazero_amd.distributed import all_gather_samples, train_step
            n_s = args.games * 64
            g = torch.Generator(device=coll_dev).manual_seed(rank)
            s_states = (torch.rand((n_s, 3, 8, 8), device=coll_dev, generator=g) > 0.7).float()
            s_pis = torch.softmax(torch.rand((n_s, 192), device=coll_dev, generator=g), 1)
            s_wls = torch.tensor([[1.0, 0.0]], device=coll_dev).repeat(n_s, 1)
            barrier()
            t1 = time.perf_counter()
            G = all_gather_samples(s_states, s_pis, s_wls)
            torch.cuda.synchronize()
            t_gather = time.perf_counter() - t1
            train_model = AlphaZeroNet(num_blocks=args.blocks, num_filters=128).to(coll_dev)
            train_model.load_state_dict(model.state_dict())
            opt = torch.optim.Adam(train_model.parameters(), lr=1e-3, weight_decay=1e-4)
            per = max(1, 1024 // world)
            lo = rank * per
            batch = (G[0][lo:lo + per], G[1][lo:lo + per], G[2][lo:lo + per])
            del G
            train_step(train_model, opt, *batch)          # warm-up (allocator, RCCL channels)
            barrier()
            t1 = time.perf_counter()
            train_step(train_model, opt, *batch)
            torch.cuda.synchronize()
            t_train = time.perf_counter() - t1
            tt = torch.tensor([t_gather, t_train], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            exchange = {"sample_allgather_ms": float(tt[0]) * 1e3, "samples_per_rank": n_s,
                        "bytes_per_rank": n_s * 1544, "train_step_ms": float(tt[1]) * 1e3,
                        "train_global_batch": per * world, "backend": backend}
        except Exception as exc:  # the self-play line must still be printed
            exchange = {"error": repr(exc)}

    if rank == 0:
        flop_launch = tower_flop_per_position(args.blocks) * args.games
        # HBM-side bytes per tower launch come from separate rocprofv3 --pmc passes
        # (profiles/): counters cannot be read from inside this process
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_k_tower_pmc.json")
        if os.path.exists(pmc) and args.games == 1024 and args.blocks == 20:
            traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
        achieved = flop_launch / (tower_ms * 1e-3) if tower_ms > 0 else 0.0
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
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_DENSE, "traffic": traffic,
                         "kernel": "k_tower" if args.games > 512 else ("k_tower_split<2>" if args.games > 256 else "k_tower_split<1>"),
                         "launch_ms": tower_ms, "launch_samples": tower_samples,
                         "flop_per_launch": flop_launch,
                         "mean_search_depth": eng.mean_depth()},
        }
        if exchange is not None:
            out["exchange"] = exchange
        if world == 1 and args.cpu_seconds > 0:
            cores = min(16, os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(128, args.cpu_seconds, args.blocks, cores)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

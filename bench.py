"""Self-play throughput benchmark (BASELINE.json: MCTS simulations/sec and
self-play positions/sec).

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE config 4, per GPU): 1024 concurrent Breakthrough games,
800 simulations per move, Dirichlet noise on, SE-ResNet 20x128 with random-init
weights, tree reuse, finished games restart so the batch stays full.  One step
= one ply of every game = 1024 x 800 simulations.  For N > 1 the games are
sharded (one process per GPU, rank-offset seeds, no data-path collective);
started without a launcher (WORLD_SIZE unset) the script starts its own N ranks
with torch.distributed.run before it touches a GPU and relays rank 0's line.

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (the fused
SE-ResNet tower) against the dense bf16 MFMA peak using HIP-event timings taken
inside the timed region.  `cpu_baseline` times the CPU oracle (C tree + float32
torch forward) as P independent single-thread processes x 128 games playing
the same self-play loop on the host cores, before the GPU is touched.
For N > 1 the CPU leg runs once per job: in the parent when the script starts its
own ranks, on rank 0 before it joins the process group when a launcher started them.
`secondary` (N = 1) times the move-generation kernel (BASELINE config 2) and
the batch-4096 network forward (config 3) with HIP events after the timed
region, then plays one whole episode batch (every game to its end, the batch
shrinking as in the reference's execute_parallel_episodes: seconds, sims/s,
positions/s, games/s) and a stretch of continuous play with the games spread over
all stages of a game; none of it enters `value`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POSITION = 778_732_032          # whole 20x128 network, 2*MAC
PEAK_BF16_DENSE = 2.5e15
PEAK_HBM = 8.0e12
MOVEGEN_BYTES_PER_POSITION = 42          # 17 in (2 x u64 + turn) + 24 legal3 + 1 status


def tower_flop_per_position(blocks):
    """FLOPs (2*MAC) of what the fused tower kernel computes for one position:
    stem conv + blocks x (two 3x3 convs + SE) + policy conv + value conv.
    20 blocks: 775,061,504."""
    conv = 2 * 64 * 128 * 128 * 9
    return 2 * 64 * 128 * 27 + blocks * (2 * conv + 12_288) + conv + 2 * 64 * 32 * 128


def network_flop_per_position(blocks):
    """Tower + the three head FCs (policy 8192x192, value 2048x128, 128x2)."""
    return tower_flop_per_position(blocks) + 2 * (8192 * 192 + 2048 * 128 + 128 * 2)


# ----------------------------------------------------------------------------
# CPU baseline: the reference's execute_parallel_episodes (src/parallel_trainer.py:137-223)
# restated over the C oracle, one single-thread process per host core.

def cpu_worker(games, sims, blocks, seconds, seed):
    """One process: `games` games in lock-step, `sims` simulations per move, root noise,
    temperature schedule, finished games leave the batch; plays until `seconds` are over.
    Prints {"sims": .., "moves": .., "evals": .., "seconds": ..}."""
    from oracle import oracle as orc
    from alphazero_amd.model import AlphaZeroNet
    torch.set_num_threads(1)
    torch.manual_seed(0)
    model = AlphaZeroNet(num_blocks=blocks, num_filters=128).eval()
    m = orc.OracleMCTS(games, c_puct=1.5)
    rs = np.random.RandomState(seed)
    for g in range(games):
        m.set_game(g, orc.start_position())
    evals = 0

    def evaluate(planes, need):
        nonlocal evals
        rows = np.flatnonzero(need)
        probs = np.zeros((games, 192), np.float32)
        values = np.zeros(games, np.float32)
        if len(rows):
            with torch.no_grad():
                pl, wl = model(torch.from_numpy(planes[rows]))
                probs[rows] = torch.softmax(pl, dim=1).numpy()
                wlp = torch.softmax(wl, dim=1).numpy()
            values[rows] = wlp[:, 0] - wlp[:, 1]
            evals += len(rows)
        return probs, values

    active = np.ones(games, np.uint8)
    plies = np.zeros(games, np.int64)
    sims_done = moves = 0
    t0 = time.perf_counter()
    deadline = t0 + seconds
    while active.any() and time.perf_counter() < deadline:
        m.set_active(active)
        if m.roots_prepare():
            probs, _ = evaluate(m.planes, m.need)
            m.roots_expand(probs)
        m.add_noise(rs.dirichlet([0.35] * 48, size=games), 0.25)
        n_active = int(active.sum())
        done = 0
        while done < sims and time.perf_counter() < deadline:
            m.select()
            probs, values = evaluate(m.planes, m.need)
            m.expand_backup(probs, values)
            sims_done += n_active
            done += 1
        if done < sims:
            break
        for g in np.flatnonzero(active):
            probs = m.action_probs(int(g), 1.0 if plies[g] < 16 else 0.0).astype(np.float64)
            a = int(rs.choice(192, p=probs / probs.sum()))
            m.advance(int(g), a)
            plies[g] += 1
            moves += 1
            if orc.is_terminal(m.get_game(int(g))):
                active[g] = 0
    print(json.dumps({"sims": sims_done, "moves": moves, "evals": evals,
                      "seconds": time.perf_counter() - t0}), flush=True)


def host_cores(world=1):
    """Cores this job may use: the affinity mask, capped by the pool's per-job share when the
    environment states one (OMP_NUM_THREADS is set to it on the GPU boxes, where the
    affinity mask still shows the whole machine).  torch.distributed.run sets OMP_NUM_THREADS=1
    for its workers when nobody else set it: that is no statement about the job's share, and the
    cap is then the boxes' 16 cores per GPU."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = os.environ.get("OMP_NUM_THREADS", "")
    injected = share == "1" and "TORCHELASTIC_RUN_ID" in os.environ
    if injected:
        n = min(n, 16 * max(1, world))
    elif share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(sims, blocks, seconds, games=128, procs=None, world=1):
    """P = host cores independent processes x 128 games, one torch thread each.  Runs before
    this process touches the GPU."""
    P = procs or host_cores(world)
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(i), "--sims", str(sims),
                               "--blocks", str(blocks), "--cpu-seconds", str(seconds), "--games", str(games)],
                              stdout=subprocess.PIPE, env=env, cwd=ROOT) for i in range(P)]
    rows = []
    for p in procs:
        out, _ = p.communicate()
        if p.returncode == 0 and out.strip():
            rows.append(json.loads(out.decode().strip().splitlines()[-1]))
    wall = time.perf_counter() - t0
    if not rows:
        return {"error": "no CPU worker finished"}
    total_sims = sum(r["sims"] for r in rows)
    span = max(r["seconds"] for r in rows)
    moves = sum(r["moves"] for r in rows)
    return dict(value=total_sims / span, unit="sims/s", cores=len(rows), kind="port",
                positions_per_sec=moves / span if moves else total_sims / span / sims,
                positions_per_sec_is="measured moves/s" if moves else f"sims/s / {sims} (no move completed inside the sample)",
                sample=f"P = {len(rows)} independent processes (one per host core of this job's share, 1 torch thread "
                       f"each) x {games} games of the reference's self-play loop from the start position, {sims} "
                       f"sims/move, Dirichlet noise, C oracle tree + fp32 torch forward {blocks}x128, {span:.1f} s "
                       f"of play per process ({wall:.1f} s wall with start-up), {total_sims} simulations in total")


# ----------------------------------------------------------------------------
# Secondary measurements: BASELINE configs 2 and 3 (HIP events on the launch stream)

def secondary_block(blocks):
    import ctypes as C
    from alphazero_amd import _lib
    from alphazero_amd.game import movegen_encode
    from alphazero_amd.model import AlphaZeroNet, HipInference
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from movegen_bench import random_positions
    out = {}
    dev = torch.device("cuda", torch.cuda.current_device())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda x: C.c_void_p(x.data_ptr())
    base_w, base_b, base_t = random_positions(1_000_000, seed=1)
    for label, n, iters in (("movegen_1M", 1_000_000, 200), ("movegen_64M", 64_000_000, 20)):
        rep = (n + base_w.numel() - 1) // base_w.numel()
        w, b, t = (x.repeat(rep)[:n].contiguous() for x in (base_w, base_b, base_t))
        legal3 = torch.empty((n, 3), dtype=torch.int64, device=dev)
        status = torch.empty(n, dtype=torch.int8, device=dev)

        def run():
            _lib.check(_lib.lib().az_movegen_encode(n, p(w), p(b), p(t), p(legal3), None, None, p(status), st))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / iters
        gbs = n * MOVEGEN_BYTES_PER_POSITION / sec
        out[label] = {"kernel": "movegen_kernel<false,false>", "positions": n, "us_per_launch": sec * 1e6,
                      "positions_per_sec": n / sec, "bytes_per_position": MOVEGEN_BYTES_PER_POSITION,
                      "achieved_GBps": gbs / 1e9, "peak_GBps": PEAK_HBM / 1e9, "frac": gbs / PEAK_HBM,
                      "launches_timed": iters,
                      "note": "1 M positions = 42 MB sit inside the 256 MB Infinity Cache; the 64 M figure is the HBM one"
                      if n <= 4_000_000 else "2.7 GB working set: HBM-resident"}
        del w, b, t, legal3, status
    # config 3: batch-4096 forward on 4096 of those positions
    B = 4096
    r = movegen_encode(base_w[:B].cpu().numpy().view(np.uint64), base_b[:B].cpu().numpy().view(np.uint64),
                       base_t[:B].cpu().numpy(), want_mask=False, want_planes=True)
    planes = r["planes"].cpu().numpy().reshape(B, 3, 64) > 0.5
    weights = np.uint64(1) << np.arange(64, dtype=np.uint64)
    mine = torch.from_numpy((planes[:, 0].astype(np.uint64) * weights).sum(1, dtype=np.uint64).view(np.int64)).to(dev)
    theirs = torch.from_numpy((planes[:, 1].astype(np.uint64) * weights).sum(1, dtype=np.uint64).view(np.int64)).to(dev)
    torch.manual_seed(0)
    inf = HipInference(AlphaZeroNet(num_blocks=blocks, num_filters=128).eval(), max_batch=B)
    for _ in range(3):
        inf.forward_bits(mine, theirs)
    torch.cuda.synchronize()
    iters = 30
    inf.set_timing(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        inf.forward_bits(mine, theirs)
    e1.record()
    torch.cuda.synchronize()
    tower_ms, tower_n = inf.tower_time()
    sec = e0.elapsed_time(e1) * 1e-3 / iters
    flop = network_flop_per_position(blocks) * B
    out["forward_4096"] = {"kernels": "k_tower16 + k_head_gemm + k_head_finish", "batch": B, "ms_per_forward": sec * 1e3,
                           "positions_per_sec": B / sec, "flop_per_position": network_flop_per_position(blocks),
                           "achieved_TFLOPs": flop / sec / 1e12, "peak_TFLOPs": PEAK_BF16_DENSE / 1e12,
                           "frac": flop / sec / PEAK_BF16_DENSE, "tower_ms": tower_ms, "tower_samples": tower_n,
                           "tower_frac": (tower_flop_per_position(blocks) * B / (tower_ms * 1e-3) / PEAK_BF16_DENSE) if tower_ms else None,
                           "forwards_timed": iters}
    inf.close()
    # the same module through stock PyTorch-ROCm (MIOpen / hipBLASLt) on this GPU, as the reference would run it
    # on a GPU device (src/mcts.py:245-253: float32 planes in, eval mode, no_grad, softmaxes): float32 and, for
    # scale, bf16 autocast.  A reported baseline beside the HIP figure, not part of any product path.
    try:
        torch.manual_seed(0)
        net = AlphaZeroNet(num_blocks=blocks, num_filters=128).eval().to(dev)
        x = r["planes"].to(dev)
        res = {}
        searched = torch.backends.cudnn.benchmark
        for label, dtype, search in (("float32", None, False), ("bf16_autocast", torch.bfloat16, False),
                                     ("float32_solver_search", None, True), ("bf16_autocast_solver_search", torch.bfloat16, True)):
            torch.backends.cudnn.benchmark = search       # True: MIOpen times its solvers per layer shape first
            def fwd():
                with torch.no_grad():
                    if dtype is None:
                        pl, wl = net(x)
                    else:
                        with torch.autocast(device_type="cuda", dtype=dtype):
                            pl, wl = net(x)
                    return torch.softmax(pl.float(), 1), torch.softmax(wl.float(), 1)
            for _ in range(3):
                fwd()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fwd()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            res[label] = {"ms_per_forward": ms, "positions_per_sec": B / (ms * 1e-3),
                          "achieved_TFLOPs": flop / (ms * 1e-3) / 1e12}
        torch.backends.cudnn.benchmark = searched
        best = min(v["ms_per_forward"] for v in res.values())
        res["hip_speedup_over_float32"] = res["float32"]["ms_per_forward"] / (sec * 1e3)
        res["hip_speedup_over_the_fastest_of_these"] = best / (sec * 1e3)
        out["forward_4096"]["torch_rocm_same_gpu"] = res
        del net, x
    except Exception as exc:
        out["forward_4096"]["torch_rocm_same_gpu"] = {"error": repr(exc)}
    # single-game search as the reference's arena and web front end run it (src/arena.py:248-265, src/web.py:447-470:
    # MCTS.search on a fresh root, 200 simulations, no noise) through the drop-in MCTS: the exact search
    # (leaf_batch 1) and the opt-in leaf-parallel mode, host loop included
    try:
        from alphazero_amd.game import BreakthroughGame
        from alphazero_amd.mcts import MCTS
        torch.manual_seed(0)
        lat_model = AlphaZeroNet(num_blocks=blocks, num_filters=128).to(dev).eval()
        lat = {}
        for k in (1, 16):
            m = MCTS(lat_model, num_simulations=200, device=str(dev), leaf_batch=k)
            m.search(BreakthroughGame())                   # warm-up: engine, weight upload
            torch.cuda.synchronize()
            times = []
            for _ in range(5):
                t0 = time.perf_counter()
                root = m.search(BreakthroughGame(), add_noise=False)
                _ = max(root.children.values(), key=lambda c: c.visit_count)
                times.append(time.perf_counter() - t0)
            lat[f"leaf_batch_{k}"] = {"ms_per_move": float(np.median(times)) * 1e3, "simulations": 200,
                                      "us_per_simulation": float(np.median(times)) / 200 * 1e6}
        lat["what"] = ("MCTS.search(start position, add_noise=False), 200 simulations, 20x128, median of 5; leaf_batch 1 is "
                       "the reference's search bit for bit, leaf_batch 16 the opt-in leaf-parallel mode (other visit counts)")
        out["latency_search"] = lat
        del lat_model
    except Exception as exc:
        out["latency_search"] = {"error": repr(exc)}
    return out


def tower_kernel_name(rows):
    """The tower kernel az_net_forward launches for this many rows (csrc/net.hip, forward_partial)."""
    names = {"1": "k_tower_split<1>", "2": "k_tower_split<2>", "4": "k_tower", "8": "k_tower_split<2,8>", "16": "k_tower16"}
    forced = os.environ.get("AZ_TOWER_FORM", "0")
    if forced in names:
        return names[forced]
    return "k_tower16" if rows > 512 else ("k_tower_split<2>" if rows > 256 else "k_tower_split<1>")


def whole_game_block(args):
    """What the timed region (opening plies only) does not show: games played to the end.
    episode: the reference's execute_parallel_episodes (src/parallel_trainer.py:158-217) at the
    benchmark size, every game to its end, finished games leave the batch.  sustained: continuous
    play (a finished game restarts at once) once the games are spread over all stages of a game:
    before the clock starts game g plays g-dependent 0..55 opening plies at 16 simulations a move."""
    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import SelfPlay
    torch.manual_seed(0)
    sp = SelfPlay(AlphaZeroNet(num_blocks=args.blocks, num_filters=128).eval(), n_games=args.games,
                  num_simulations=args.sims, seed=4321)
    e = sp.engine
    out = {}
    # ---- one episode batch
    e.reset_all()
    e.set_active(np.ones(args.games, np.uint8))
    torch.cuda.synchronize()
    st0 = e.stats()
    t0 = time.perf_counter()
    plies = 0
    while e.stats()["games_finished"] - st0["games_finished"] < args.games:
        e.play_move(args.sims, add_noise=True, restart_finished=False)
        plies += 1
        if e.samples_pending() > args.games * 64:
            e.drain_samples()
    e.drain_samples()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st1 = e.stats()
    d = {k: st1[k] - st0[k] for k in ("simulations", "moves", "games_finished", "nn_evals")}
    out["episode"] = {"what": f"{args.games} games x {args.sims} sims/move played to the end, finished games leave the batch "
                              "(the reference's execute_parallel_episodes); samples drained on the device inside the timed region",
                      "seconds": dt, "plies_of_the_longest_game": plies, "games": d["games_finished"], "positions": d["moves"],
                      "simulations": d["simulations"], "sims_per_sec": d["simulations"] / dt,
                      "positions_per_sec": d["moves"] / dt, "games_per_sec": d["games_finished"] / dt}
    # ---- continuous play in the steady state
    e.reset_all()
    rs = np.random.RandomState(7)
    offset = rs.randint(0, 56, size=args.games)
    for j in range(int(offset.max())):
        e.set_active((offset > j).astype(np.uint8))
        e.play_move(16, add_noise=True, restart_finished=True)
    e.set_active(np.ones(args.games, np.uint8))
    e.drain_samples()
    torch.cuda.synchronize()
    st0 = e.stats()
    t0 = time.perf_counter()
    for i in range(args.sustained_plies):
        e.play_move(args.sims, add_noise=True, restart_finished=True)
        if (i + 1) % 16 == 0 and e.samples_pending() > 32768:
            e.drain_samples()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st1 = e.stats()
    d = {k: st1[k] - st0[k] for k in ("simulations", "moves", "games_finished", "nn_evals")}
    out["sustained"] = {"what": f"{args.sustained_plies} plies of continuous play, {args.games} games x {args.sims} sims/move, finished "
                                "games restart at once; before the clock starts the games are spread over the stages of a game "
                                "(game g plays 0..55 opening plies at 16 sims/move)",
                        "seconds": dt, "plies": args.sustained_plies, "games_finished": d["games_finished"],
                        "positions": d["moves"], "simulations": d["simulations"], "sims_per_sec": d["simulations"] / dt,
                        "positions_per_sec": d["moves"] / dt, "games_per_sec": d["games_finished"] / dt,
                        "terminal_leaf_share": 1.0 - d["nn_evals"] / max(1, d["simulations"] + d["moves"])}
    del sp, e
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """N > 1 without a launcher: start the N ranks as child processes of this one, which has
    made no GPU call, relay rank 0's JSON line and exit with the launcher's status."""
    # the CPU leg runs here, on a host that is otherwise idle, before any rank exists
    cpu = None
    if args.cpu_seconds > 0 and not args.exchange_only:
        cpu = cpu_baseline(args.sims, args.blocks, args.cpu_seconds, world=args.gpus)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["AZ_BENCH_CPU_LEG_DONE"] = "1"      # the ranks skip it
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, cwd=os.getcwd())
    line = None
    for raw in proc.stdout:
        text = raw.decode(errors="replace")
        if text.lstrip().startswith('{"metric"'):
            line = text.strip()
        else:
            sys.stderr.write(text)
    rc = proc.wait()
    if line is not None:
        if cpu is not None:
            merged = json.loads(line)
            merged["cpu_baseline"] = cpu
            line = json.dumps(merged)
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: the ranks exited without printing a result line\n")
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=1024, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800, help="simulations per move")
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config 2 / config 3 measurements and the whole-game block")
    ap.add_argument("--no-whole-games", action="store_true", help="skip secondary.episode / secondary.sustained (about a minute)")
    ap.add_argument("--sustained-plies", type=int, default=48)
    ap.add_argument("--exchange-only", action="store_true",
                    help="N > 1: time only the two exchange steps (sample all-gather, data-parallel training "
                         "step); no self-play, value = 0.  With AZ_BENCH_BACKEND=gloo this needs no GPU")
    ap.add_argument("--cpu-worker", type=int, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_worker is not None:
        cpu_worker(args.games, args.sims, args.blocks, args.cpu_seconds, 1000 + args.cpu_worker)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # CPU leg first: its worker processes are started while this process has made no GPU call
    # (N > 1 under a launcher: rank 0 runs it while the other ranks wait in the rendezvous, idle)
    cpu = None
    if rank == 0 and args.cpu_seconds > 0 and not args.exchange_only and not os.environ.get("AZ_BENCH_CPU_LEG_DONE"):
        cpu = cpu_baseline(args.sims, args.blocks, args.cpu_seconds, world=world)

    backend = os.environ.get("AZ_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not (args.exchange_only and rehearsal):
        raise SystemExit("bench.py needs a GPU: the self-play engine has no CPU fallback")
    # AZ_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with one GPU: every
    # rank then shares cuda:0 and the collectives run on CPU tensors.
    if have_gpu:
        torch.cuda.set_device(0 if rehearsal else local_rank)
    coll_dev = "cpu" if rehearsal else "cuda"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        limit = datetime.timedelta(seconds=300)    # a rank that died must not leave the others waiting for ever
        if rehearsal:
            dist.init_process_group(backend, timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)

    from alphazero_amd.model import AlphaZeroNet
    from alphazero_amd.selfplay import SelfPlay

    torch.manual_seed(0)                       # same random-init weights on every rank
    model = AlphaZeroNet(num_blocks=args.blocks, num_filters=128).eval()

    def barrier():
        if have_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if have_gpu:
            torch.cuda.synchronize()

    if args.exchange_only:
        if dist is None:
            raise SystemExit("--exchange-only needs --gpus N with N > 1")
        sp = eng = inf = None
        st0 = st1 = dict(simulations=0, moves=0, nn_evals=0)
        dt, tower_ms, tower_samples, depth = 1.0, 0.0, 0, 0.0
    else:
        sp = SelfPlay(model, n_games=args.games, num_simulations=args.sims, seed=1234 + rank)
        eng, inf = sp.engine, sp.inference
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
        barrier()
        dt = time.perf_counter() - t0
        tower_ms, tower_samples = inf.tower_time()
        inf.set_timing(False)
        st1 = eng.stats()
        depth = eng.mean_depth()
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
        exchange = exchange_block(args, dist, model, rank, world, coll_dev, backend, barrier)

    if rank == 0:
        # rows per tower launch: the launches of the timed region carried exactly the evaluated
        # leaves (terminal leaves take no row), `evals` rows over steps x (sims + 1) launches per rank
        launches = float(args.steps) * (args.sims + 1) * world
        rows_per_launch = evals / launches if launches else 0.0
        flop_launch = tower_flop_per_position(args.blocks) * rows_per_launch
        # HBM-side bytes per tower launch come from separate rocprofv3 --pmc passes
        # (profiles/): counters cannot be read from inside this process
        traffic, traffic_source = None, None
        for name in ("r03_k_tower16_pmc.json",):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc) and args.games == 1024 and args.blocks == 20:
                traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
                traffic_source = f"profiles/{name}: separate rocprofv3 --pmc passes of this workload, not this run"
                break
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
            "data": "synthetic (random-init weights, games from the start position; the timed region covers opening plies, "
                    "secondary.sustained / secondary.episode cover whole games)",
            "config": {"workload": f"full self-play: {args.games} concurrent games/GPU x {args.sims} sims/move, "
                                   f"Dirichlet noise, SE-ResNet {args.blocks}x128, tree reuse, restart on finish",
                       "games_per_gpu": args.games, "sims_per_move": args.sims,
                       "parallelism": f"games sharded over {world} GPU(s), no data-path collective"},
            "roofline": {"bound": "mfma", "achieved": achieved / 1e12, "peak": PEAK_BF16_DENSE / 1e12,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_DENSE, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel": tower_kernel_name(args.games),
                         "launch_ms": tower_ms, "launch_samples": tower_samples,
                         "flop_per_launch": flop_launch,
                         "rows_per_launch": rows_per_launch,
                         "mean_search_depth": depth,
                         # the whole job against the same peak: evaluated positions/s x FLOP per position of the
                         # whole network (tower + heads), tree kernels, head GEMM and launch gaps included in the time
                         "end_to_end_achieved": evals / dt * network_flop_per_position(args.blocks) / 1e12,
                         "end_to_end_frac": evals / dt * network_flop_per_position(args.blocks) / (PEAK_BF16_DENSE * world)},
        }
        if exchange is not None:
            out["exchange"] = exchange
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if args.exchange_only:
            out["config"]["workload"] = "exchange steps only (no self-play was run; value is 0 by construction)"
        if world == 1 and not args.no_secondary:
            del sp, eng, inf
            torch.cuda.empty_cache()
            try:
                out["secondary"] = secondary_block(args.blocks)
            except Exception as exc:           # the headline line must still be printed
                out["secondary"] = {"error": repr(exc)}
            if not args.no_whole_games:
                try:
                    out["secondary"].update(whole_game_block(args))
                    out["games_per_sec"] = out["secondary"]["sustained"]["games_per_sec"]
                    out["games_per_sec_is"] = "secondary.sustained (continuous play in the steady state; the timed region " \
                                              "plays opening plies, where no game ends); secondary.episode has the " \
                                              "reference's shrinking-batch figure"
                except Exception as exc:
                    out["secondary"]["whole_games_error"] = repr(exc)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def exchange_block(args, dist, model, rank, world, coll_dev, backend, barrier):
    """The two exchange steps of an iteration, timed outside the self-play region on a
    synthetic iteration's worth of samples (games x 64 plies per rank): all-gather into
    every rank's replay buffer, then one data-parallel training step at global batch 1024
    with a gradient all-reduce (RCCL when backend is nccl)."""
    from alphazero_amd.model import AlphaZeroNet

    def all_ok(ok):
        """True when every rank got here without an error: a rank that failed on its own must not
        leave the others waiting inside the next collective."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    state, err = {}, None

    def stage(fn):
        """Runs fn (local work only) on this rank unless an earlier stage failed somewhere; then all
        ranks agree on whether to go on.  The collectives themselves sit between stages."""
        nonlocal err
        if err is None:
            try:
                fn()
            except Exception as exc:
                err = repr(exc)
        if not all_ok(err is None):
            err = err or "another rank failed in this stage"
            return False
        return True

    def prepare():
        from alphazero_amd.distributed import all_gather_packed, train_step  # noqa: F401
        from alphazero_amd._lib import PACKED_SAMPLE_BYTES
        n_s = args.games * 64
        g = torch.Generator(device=coll_dev).manual_seed(rank)
        # packed rows as the engine queues them (bitboards | wl | pi); the contents do not matter here
        packed = torch.zeros((n_s, PACKED_SAMPLE_BYTES), dtype=torch.uint8, device=coll_dev)
        fl = packed[:, 16:].view(torch.float32)
        fl[:, 0] = 1.0
        fl[:, 2:] = torch.softmax(torch.rand((n_s, 192), device=coll_dev, generator=g), 1)
        packed[:, :16] = torch.randint(0, 2, (n_s, 16), device=coll_dev, generator=g, dtype=torch.uint8)
        state.update(packed=packed, n_s=n_s, row=PACKED_SAMPLE_BYTES)

    def unpack():
        gathered = state.pop("gathered")
        if gathered.is_cuda:
            from alphazero_amd.engine import unpack_samples
            t1 = time.perf_counter()
            G = unpack_samples(gathered)
            torch.cuda.synchronize()
            state["t_unpack"] = time.perf_counter() - t1
        else:                                         # gloo rehearsal: the unpack kernel needs the rows on a GPU
            fl = gathered[:, 16:].view(torch.float32)
            G = (torch.zeros((len(gathered), 3, 8, 8)), fl[:, 2:].clone(), fl[:, :2].clone())
            G[0][:, 2] = 1.0
            state["t_unpack"] = None
        train_model = AlphaZeroNet(num_blocks=args.blocks, num_filters=128).to(coll_dev)
        train_model.load_state_dict(model.state_dict())
        opt = torch.optim.Adam(train_model.parameters(), lr=1e-3, weight_decay=1e-4)
        per = max(1, 1024 // world)
        lo = rank * per
        state.update(train_model=train_model, opt=opt, per=per, batch=(G[0][lo:lo + per], G[1][lo:lo + per], G[2][lo:lo + per]))

    try:
        from alphazero_amd.distributed import all_gather_packed, train_step
        if not stage(prepare):
            return {"error": err}
        all_gather_packed(state["packed"][:16])       # warm-up (RCCL channels)
        barrier()
        t1 = time.perf_counter()
        state["gathered"] = all_gather_packed(state["packed"])
        barrier()
        t_gather = time.perf_counter() - t1
        if not stage(unpack):
            return {"error": err}
        train_step(state["train_model"], state["opt"], *state["batch"])   # warm-up (allocator, RCCL channels)
        barrier()
        t1 = time.perf_counter()
        train_step(state["train_model"], state["opt"], *state["batch"])
        barrier()
        t_train = time.perf_counter() - t1
        tt = torch.tensor([t_gather, t_train], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_unpack = state["t_unpack"]
        return {"sample_allgather_ms": float(tt[0]) * 1e3, "samples_per_rank": state["n_s"],
                "bytes_per_rank": state["n_s"] * state["row"], "wire_format": "packed rows of 792 B (bitboards | wl | pi f32)",
                "unpack_ms": None if t_unpack is None else t_unpack * 1e3,
                "train_step_ms": float(tt[1]) * 1e3,
                "train_global_batch": state["per"] * world, "backend": backend}
    except Exception as exc:  # the self-play line must still be printed
        return {"error": repr(exc)}


if __name__ == "__main__":
    main()

"""CPU checks of bench.py: the FLOP accounting it prices the tower with, and the CPU
baseline leg (oracle tree + float32 torch forward) on a tiny sample."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("az_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_flop_accounting(bench):
    conv = 2 * 64 * 128 * 128 * 9                       # one 3x3 128->128 conv on 64 squares
    stem = 2 * 64 * 128 * 27
    tower20 = stem + 20 * (2 * conv + 12_288) + conv + 2 * 64 * 32 * 128
    assert bench.tower_flop_per_position(20) == tower20 == 775_061_504
    heads = 2 * 8192 * 192 + 2 * 2048 * 128 + 2 * 128 * 2
    assert tower20 + heads == bench.FLOP_PER_POSITION == 778_732_032
    assert bench.PEAK_BF16_DENSE == 2.5e15


def test_cpu_baseline_leg_runs(bench):
    """P single-thread processes playing the reference's self-play loop over the oracle."""
    out = bench.cpu_baseline(sims=3, blocks=1, seconds=2.0, games=4, procs=2)
    assert out["kind"] == "port" and out["unit"] == "sims/s" and out["cores"] == 2
    assert out["value"] > 0 and "P = 2 independent processes" in out["sample"] and "4 games" in out["sample"]
    assert out["positions_per_sec"] > 0


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start two ranks itself and relay one
    JSON line.  Rehearsed here without a GPU: gloo backend, exchange steps only (the sample
    all-gather and the data-parallel training step are real, the self-play region is skipped)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(AZ_BENCH_BACKEND="gloo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--games", "4", "--blocks", "1", "--exchange-only"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["metric"] == "selfplay_mcts_simulations_per_sec"
    ex = out["exchange"]
    assert "error" not in ex, ex
    assert ex["backend"] == "gloo" and ex["sample_allgather_ms"] > 0 and ex["train_step_ms"] > 0
    assert ex["samples_per_rank"] == 4 * 64

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
    out = bench.cpu_baseline(games=4, sims_budget_s=0.5, blocks=1, cores=2)
    assert out["kind"] == "port" and out["unit"] == "sims/s" and out["cores"] == 2
    assert out["value"] > 0 and "4 games" in out["sample"]

#!/bin/bash
# Self-play bench and bare forward time for tower forms 16 and 4 on one box: bash tools/bench_ab.sh <name>
OUT=gpurun_out/${1:-bench_ab}.txt
mkdir -p gpurun_out
{
for f in 16 4; do
  AZ_TOWER_FORM=$f timeout -k 10 170 python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary 2>/dev/null > gpurun_out/_bench_ab.json
  python3 - "$f" <<'PY'
import json, sys
d = json.load(open("gpurun_out/_bench_ab.json"))
r = d["roofline"]
print(f"bench.py, form {sys.argv[1]}: {d['value']:.0f} sims/s, {r['kernel']} {r['launch_ms'] * 1e3:.1f} us per launch over {r['launch_samples']} samples, frac {r['frac']:.4f}, rows per launch {r['rows_per_launch']:.1f}")
PY
  AZ_TOWER_FORM=$f timeout -k 10 100 python3 tools/net_bench.py --batch 1024 --iters 300 2>/dev/null | tail -1
done
} > $OUT 2>&1
cat $OUT

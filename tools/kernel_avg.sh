#!/bin/bash
# Prints rocprofv3 average kernel durations of tools/net_bench.py (run on the GPU box).
#   bash tools/kernel_avg.sh <tag> [net_bench args]
TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kavg_$TAG -- python3 tools/net_bench.py --batch 1024 --iters 50 "$@" > gpurun_out/kavg_$TAG.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
for f in glob.glob("gpurun_out/kavg_$TAG/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "az::" in r["Name"]:
            print(f'$TAG {r["Name"].split("(")[0]:24s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.2f} us')
PY

#!/bin/bash
# rocprofv3 --kernel-trace --stats of the four measured workloads, one run each.  Run on the GPU
# box from the repo root:   bash tools/profile_stats.sh <tag>
# Leaves <tag>_<workload>_kernel_stats.csv (and the bench line of the self-play run) in gpurun_out/.
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
stats() {  # name, then the command
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1 || { echo "$name failed"; tail -5 $OUT/$name.log; return 1; }
  local f
  f=$(find $OUT/$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_${name}_kernel_stats.csv
  grep -h '^{"metric"' $OUT/$name.log > gpurun_out/${TAG}_${name}_bench_under_rocprof.json 2>/dev/null || rm -f gpurun_out/${TAG}_${name}_bench_under_rocprof.json
  tail -2 $OUT/$name.log
}
stats selfplay python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary || exit 1
AZ_TOWER_FORM=4 stats selfplay_form4 python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary || exit 1
stats movegen_1M python3 tools/movegen_bench.py --n 1000000 --iters 200 || exit 1
stats movegen_64M python3 tools/movegen_bench.py --n 64000000 --iters 20 || exit 1
stats forward_4096 python3 tools/net_bench.py --batch 4096 --iters 30 || exit 1
echo done

#!/bin/bash
# Collects rocprofv3 evidence for the hot kernels.  Run on the GPU box from the repo root:
#   bash tools/profile_pmc.sh <tag>
# Every counter pass is its own run (kernel-trace/stats never combined with --pmc).
set -u
TAG=${1:-r03}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_pass() {  # name, counters..., then -- command
  local name=$1; shift
  local ctrs=()
  while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  timeout -s KILL 120 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $OUT/$name -- "$@" > $OUT/$name.log 2>&1 || { echo "pass $name failed"; return 1; }
}
NET="python3 tools/net_bench.py --batch 1024 --iters 20"   # AZ_TOWER_FORM in the environment picks the tower kernel
MOV="python3 tools/movegen_bench.py --n 16000000 --iters 10"
run_pass tower_fetch FETCH_SIZE -- $NET || exit 1
run_pass tower_write WRITE_SIZE -- $NET || exit 1
run_pass tower_sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -- $NET || exit 1
run_pass tower_lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -- $NET || exit 1
run_pass movegen_fetch FETCH_SIZE -- $MOV || exit 1
run_pass movegen_write WRITE_SIZE -- $MOV || exit 1
run_pass selfplay_fetch FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --sims 100 --cpu-seconds 0 --no-secondary || exit 1
run_pass selfplay_write WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --sims 100 --cpu-seconds 0 --no-secondary || exit 1
echo done

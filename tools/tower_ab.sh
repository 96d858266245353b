#!/bin/bash
# A/B of tower forms on the GPU box: parity test of every form, interleaved timing, cycle stamps of the form under test.
# usage: bash tools/tower_ab.sh <log name> [form under test]
set -euo pipefail
LOG=gpurun_out/$1.log
F=${2:-16}
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_net.py -x -q --timeout 120 --timeout-method thread -k "every_tower_form" > $LOG 2>&1
timeout -k 10 300 python -u tools/tower_ab.py --batch 1024 --forms 4,$F --rounds 10 --iters 200 >> $LOG 2>&1
AZ_TOWER_STAMPS=1 AZ_TOWER_FORM=$F timeout -k 10 120 python -u tools/net_bench.py --batch 1024 --iters 2000 >> $LOG 2>&1
grep -v amdgpu.ids $LOG

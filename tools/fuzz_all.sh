#!/bin/bash
# The three differential fuzzers on the current kernels (GPU box): bash tools/fuzz_all.sh <out file under gpurun_out/>
OUT=gpurun_out/${1:-fuzz_parity.txt}
mkdir -p gpurun_out
{
echo "## python3 tests/fuzz_tree_parity.py --rounds 3000 --seed 21001   (device engine vs the C oracle)"
timeout -k 10 500 python3 -u tests/fuzz_tree_parity.py --rounds 3000 --seed 21001 2>&1 | tail -1
echo "## python3 tests/fuzz_fused_parity.py --rounds 3000 --seed 31001   (fused search, device row assignment, every tower form incl. 16, vs the step API)"
timeout -k 10 400 python3 -u tests/fuzz_fused_parity.py --rounds 3000 --seed 31001 2>&1 | tail -1
echo "## python3 tests/fuzz_net_parity.py --rounds 1000 --seed 41001   (HIP forward, every tower form incl. 16, vs float32 torch)"
timeout -k 10 400 python3 -u tests/fuzz_net_parity.py --rounds 1000 --seed 41001 2>&1 | tail -1
} > $OUT 2>&1
cat $OUT

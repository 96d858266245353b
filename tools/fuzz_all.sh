#!/bin/bash
# The three differential fuzzers on the current kernels (GPU box; each writes its progress to a log under gpurun_out/;
# a fuzzer that fails ends the script: nothing more is started on the GPU behind it):
#   bash tools/fuzz_all.sh <out file under gpurun_out/> [tree rounds] [fused rounds] [net rounds] [seed offset]
OUT=gpurun_out/${1:-fuzz_parity.txt}
TR=${2:-3000}; FR=${3:-3000}; NR=${4:-1000}; SO=${5:-0}
S1=$((21001 + SO)); S2=$((31001 + SO)); S3=$((41001 + SO))
mkdir -p gpurun_out
{
echo "## python3 tests/fuzz_tree_parity.py --rounds $TR --seed $S1   (device engine vs the C oracle)"
timeout -k 10 1000 python3 -u tests/fuzz_tree_parity.py --rounds $TR --seed $S1 > gpurun_out/fuzz_tree.log 2>&1 || { tail -5 gpurun_out/fuzz_tree.log; echo "stopped: the tree fuzzer failed"; exit 1; }; tail -1 gpurun_out/fuzz_tree.log
echo "## python3 tests/fuzz_fused_parity.py --rounds $FR --seed $S2   (fused search, device row assignment, every tower form incl. 16, vs the step API)"
timeout -k 10 600 python3 -u tests/fuzz_fused_parity.py --rounds $FR --seed $S2 > gpurun_out/fuzz_fused.log 2>&1 || { tail -5 gpurun_out/fuzz_fused.log; echo "stopped: the fused fuzzer failed"; exit 1; }; tail -1 gpurun_out/fuzz_fused.log
echo "## python3 tests/fuzz_net_parity.py --rounds $NR --seed $S3   (HIP forward, every tower form incl. 16 and, in a quarter of the rounds, another width, vs float32 torch)"
timeout -k 10 600 python3 -u tests/fuzz_net_parity.py --rounds $NR --seed $S3 > gpurun_out/fuzz_net.log 2>&1 || { tail -5 gpurun_out/fuzz_net.log; echo "stopped: the network fuzzer failed"; exit 1; }; tail -1 gpurun_out/fuzz_net.log
} > $OUT 2>&1
cat $OUT

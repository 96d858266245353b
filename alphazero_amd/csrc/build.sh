#!/bin/bash
# Builds alphazero_amd/libazamd.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libazamd.so
OBJ=../../build/obj
mkdir -p "$OBJ"
COMMON="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-result"
pids=()
# tree/movegen: exact float32 semantics -> no FMA contraction
$HIPCC $COMMON -ffp-contract=off -c tree.hip -o $OBJ/tree.o & pids+=($!)
$HIPCC $COMMON -ffp-contract=off -c movegen.hip -o $OBJ/movegen.o & pids+=($!)
$HIPCC $COMMON -ffp-contract=off -c selfplay.hip -o $OBJ/selfplay.o & pids+=($!)
$HIPCC $COMMON -ffp-contract=off -c leafpar.hip -o $OBJ/leafpar.o & pids+=($!)
# net: accumulators in architectural VGPRs (no v_accvgpr_read per element in the epilogues)
$HIPCC $COMMON -mllvm -amdgpu-mfma-vgpr-form=1 -DAZ_SPLIT_STAMPS=${AZ_SPLIT_STAMPS:-0} -c net.hip -o $OBJ/net.o & pids+=($!)
# tower16: the compiler's default register form (accumulators in AGPRs).  With the VGPR form the 128
# accumulators, 96 fragment registers, the packed residual and the tap addresses exceed 256 registers and
# the allocator moves accumulators in and out of AGPRs inside the tap loop (s_nop 7 after the MFMA each time).
$HIPCC $COMMON ${T16_FLAGS-} -Rpass-analysis=kernel-resource-usage -c tower16.hip -o $OBJ/tower16.o 2> $OBJ/tower16.remarks & pids+=($!)
$HIPCC $COMMON -c tower_any.hip -o $OBJ/tower_any.o & pids+=($!)
g++ -O2 -fPIC -std=c++17 -c core.cpp -o $OBJ/core.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/*.o
echo "built $(readlink -f $OUT)"

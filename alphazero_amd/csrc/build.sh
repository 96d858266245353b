#!/bin/bash
# Builds alphazero_amd/libazamd.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libazamd.so
OBJ=../../build/obj
COMMON="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-result"
# HOSTSAN=1: the same library with the HOST code under UndefinedBehaviorSanitizer in trap mode (no
# runtime to link or preload: undefined behaviour in the host code ends the process with SIGILL), the
# device code unchanged.  Run the GPU suite on it with AZAMD_LIB=alphazero_amd/libazamd_hostsan.so.
if [ "${HOSTSAN:-0}" = "1" ]; then
  OUT=../libazamd_hostsan.so
  OBJ=../../build/obj_hostsan
  COMMON="$COMMON -Xarch_host -fsanitize=undefined -Xarch_host -fsanitize-trap=undefined"
fi
mkdir -p "$OBJ"
pids=()
# tree/movegen: exact float32 semantics -> no FMA contraction
$HIPCC $COMMON -ffp-contract=off -c tree.hip -o $OBJ/tree.o & pids+=($!)
$HIPCC $COMMON -ffp-contract=off -c movegen.hip -o $OBJ/movegen.o & pids+=($!)
$HIPCC $COMMON -ffp-contract=off -c selfplay.hip -o $OBJ/selfplay.o & pids+=($!)
$HIPCC $COMMON -ffp-contract=off -c leafpar.hip -o $OBJ/leafpar.o & pids+=($!)
# net: accumulators in architectural VGPRs (no v_accvgpr_read per element in the epilogues)
$HIPCC $COMMON -mllvm -amdgpu-mfma-vgpr-form=1 -DAZ_SPLIT_STAMPS=${AZ_SPLIT_STAMPS:-0} -c net.hip -o $OBJ/net.o & pids+=($!)
# tower16: accumulators in architectural VGPRs as well.  The form is a measured choice (T16_FLAGS="" gives the
# compiler's default, accumulators in AGPRs): 128 accumulators + 96 fragment registers + 64 of packed residual
# + 36 tap addresses exceed 256 VGPRs.  While the conv loop ran over single convs the allocator answered by moving
# accumulators through AGPRs inside the tap loop (s_nop 7 behind the MFMA each time) and the AGPR form was 2 %
# faster; with the loop over blocks (literal conv parity, no phi copies) it parks the residual and the tap
# addresses in AGPRs instead, the loops are clean in both forms and the VGPR form saves the 128 accumulator reads
# of every epilogue: 0.877 against 0.886 of k_tower's time per forward.  Check the tap loop's instruction mix
# (v_accvgpr_*, s_nop 7) after any change to tower16.hip: python tools/tower16_isa_check.py [--agpr].
$HIPCC $COMMON ${T16_FLAGS--mllvm -amdgpu-mfma-vgpr-form=1} -Rpass-analysis=kernel-resource-usage -c tower16.hip -o $OBJ/tower16.o 2> $OBJ/tower16.remarks & pids+=($!)
$HIPCC $COMMON -c tower_any.hip -o $OBJ/tower_any.o & pids+=($!)
g++ -O2 -fPIC -std=c++17 ${HOSTSAN:+-fsanitize=undefined -fsanitize-undefined-trap-on-error} -c core.cpp -o $OBJ/core.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJ/*.o
echo "built $(readlink -f $OUT)"

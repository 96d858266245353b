"""Instruction mix of k_tower16's tap loops, from a fresh device-only compile of tower16.hip with build.sh's flags.
The register allocator's treatment of this kernel changes with unrelated edits: run this after any change to it.

    python tools/tower16_isa_check.py            # the form build.sh ships (accumulators in VGPRs)
    python tools/tower16_isa_check.py --agpr     # the compiler's default form

A healthy tap loop is a basic block of 1,152 MFMAs with 2,080 to 2,150 instructions: 440 ds_read_b128, 72-79
global_load_lds_dwordx4, about 30 v_accvgpr_read_b32 (tap addresses parked in AGPRs), at most a few
v_accvgpr_write_b32, 40 to 65 s_nop cycles, and the kernel reports no spill."""
import collections
import os
import re
import subprocess
import sys
import tempfile

here = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(os.path.dirname(here), "alphazero_amd", "csrc", "tower16.hip")
flags = [] if "--agpr" in sys.argv else ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "tower16.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", *flags, "--cuda-device-only", "-S",
                    src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
blocks, cur = [], ["entry", collections.Counter()]
for line in text.split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):", line)
    if m:
        blocks.append(cur)
        cur = [m.group(1), collections.Counter()]
        continue
    m = re.match(r"^\s+([a-z_0-9]+)", line)
    if m and not line.strip().startswith((".", ";")):
        cur[1][m.group(1)] += 1
        if m.group(1) == "s_nop":
            cur[1]["nop_cycles"] += int(line.split()[1]) + 1
blocks.append(cur)
for name, c in blocks:
    mf = c["v_mfma_f32_16x16x32_bf16"]
    if mf >= 100:
        total = sum(c.values()) - c["nop_cycles"]
        print(f"{name}: {total} instructions, {mf} MFMA, {c['ds_read_b128']} ds_read_b128, {c['global_load_lds_dwordx4']} LDS-DMA copies, "
              f"v_accvgpr read/write/mov {c['v_accvgpr_read_b32']}/{c['v_accvgpr_write_b32']}/{c['v_accvgpr_mov_b32']}, "
              f"s_nop cycles {c['nop_cycles']}, scratch ops {sum(v for k, v in c.items() if k.startswith('scratch_'))}")
for key in ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
    m = re.search(rf"\.{key}:\s+(\d+)", text)
    if m:
        print(f"{key}: {m.group(1)}")

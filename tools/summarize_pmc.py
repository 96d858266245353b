"""Summarises rocprofv3 --pmc CSVs (one directory per pass) into per-kernel means."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    name = os.path.basename(d.rstrip("/"))
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
    print(f"## pass {name}")
    for (kern, ctr), (tot, n) in sorted(acc.items()):
        if n >= 1 and ("az::" in kern):
            print(f"{kern:28s} {ctr:32s} mean/dispatch {tot / n:16.1f}  dispatches {n}")

# tower summary as JSON (second argument: output path; third: kernel, default az::k_tower16), in the units bench.py reads:
# FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 the fetch counter counts the 128-byte
# requests of 16-byte-per-lane streams as 64 bytes, hence the x2 on the fetch side
if len(sys.argv) > 2:
    import json

    def mean(pass_name, kernel, counter):
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(root, pass_name, "*", "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if row["Kernel_Name"].split("(")[0] == kernel and row["Counter_Name"] == counter:
                    tot += float(row["Counter_Value"])
                    n += 1
        return tot / n if n else None
    k = sys.argv[3] if len(sys.argv) > 3 else "az::k_tower16"
    fetch, write = mean("tower_fetch", k, "FETCH_SIZE"), mean("tower_write", k, "WRITE_SIZE")
    busy, gui = mean("tower_sq", k, "SQ_VALU_MFMA_BUSY_CYCLES"), mean("tower_sq", k, "GRBM_GUI_ACTIVE")
    out = {"kernel": k.split("::")[-1], "batch": 1024,
           "fetch_bytes_reported": fetch * 1024, "fetch_bytes_corrected_x2": fetch * 2048, "write_bytes": write * 1024,
           "traffic_bytes_per_launch": fetch * 2048 + write * 1024,
           "mfma_busy_cycles": busy, "grbm_gui_active": gui,
           "mfma_pipe_busy_frac": busy / 1024 / (gui / 8) if busy and gui else None,
           "sq_wave_cycles": mean("tower_sq", k, "SQ_WAVE_CYCLES"), "sq_busy_cycles": mean("tower_sq", k, "SQ_BUSY_CYCLES"),
           "mfma_mops_bf16": mean("tower_sq", k, "SQ_INSTS_VALU_MFMA_MOPS_BF16"),
           "lds_insts": mean("tower_lds", k, "SQ_INSTS_LDS"),
           "lds_bank_conflict_cycles": mean("tower_lds", k, "SQ_LDS_BANK_CONFLICT"),
           "lds_idx_active_cycles": mean("tower_lds", k, "SQ_LDS_IDX_ACTIVE"),
           "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, separate passes, "
                     "tools/profile_pmc.sh; tools/net_bench.py --batch 1024"}
    json.dump(out, open(sys.argv[2], "w"), indent=1)

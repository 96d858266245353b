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

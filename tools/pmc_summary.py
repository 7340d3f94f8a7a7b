#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch for kernels matching a substring.
usage: python tools/pmc_summary.py <pmc_dir> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "sparse_attn_mfma"
for sub in sorted(glob.glob(os.path.join(d, "*", "pmc_counter_collection.csv"))):
    acc = defaultdict(list)
    with open(sub) as f:
        for row in csv.DictReader(f):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    kt = os.path.join(os.path.dirname(sub), "pmc_kernel_trace.csv")
    durs = []
    if os.path.exists(kt):
        with open(kt) as f:
            for row in csv.DictReader(f):
                if pat in row["Kernel_Name"]:
                    durs.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    name = os.path.basename(os.path.dirname(sub))
    print(f"[{name}] dispatches={len(durs)} avg_us={sum(durs)/max(1,len(durs)):.1f}")
    for k, v in acc.items():
        print(f"   {k:32s} {sum(v)/len(v):16.1f}")

#!/usr/bin/env python3
"""usage: python tools/pmc_traffic_x3p_summary.py <dir with fetch/ and write/> cfgB|cfgC|cfgA [out.json]
Bytes per CALL of snf_sparse_attn_fwd_x3_hl (all its launches: Kp split, statistics / main passes, reductions), with the unit
corrections of MI355X_MICROARCH.md: both counters are calibrated on a streaming read / write of known size in the same run
(FETCH_SIZE reports half the bytes of a wide coalesced read on gfx950: the calibration finds the factor 2)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d, wl = sys.argv[1], sys.argv[2]
N, D, h, K = {"cfgB": (32768, 768, 6, 200), "cfgC": (100000, 768, 6, 512), "cfgA": (8192, 384, 6, 200)}[wl]
KNOWN = 32768 * 768 * 4
REPS = 8


def rows(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def pick(tab, sub):
    return [v for k, v in tab.items() if sub in k]


fetch, write = rows("fetch", "FETCH_SIZE"), rows("write", "WRITE_SIZE")
cal_r = pick(fetch, "critic_kernel")[0]
cal_w = pick(write, "FillFunctor")[0]
fr, fw = KNOWN / (sum(cal_r[2:]) / len(cal_r[2:])), KNOWN / (sum(cal_w[2:]) / len(cal_w[2:]))
print("bytes per counter unit: FETCH %.1f (1024 x %.2f) | WRITE %.1f (1024 x %.2f)" % (fr, fr / 1024, fw, fw / 1024))
out = {}
tot = 0
names = []
for name in ("sparse_attn_x3p_kernel", "x3p_prep_kp_kernel", "x3p_reduce_kernel"):
    if not pick(fetch, name):
        continue                   # (no prep launch when the key projection wrote the fragment image)
    names.append(name)
    rb = sum(sum(v) for v in pick(fetch, name)) * fr / REPS
    wb = sum(sum(v) for v in pick(write, name)) * fw / REPS
    nl = sum(len(v) for v in pick(fetch, name)) / REPS
    out[name] = {"launches_per_call": nl, "read_bytes": round(rb), "write_bytes": round(wb)}
    tot += rb + wb
    print("%-26s %4.1f launches / call   read %9.2f MB  write %8.2f MB per call" % (name, nl, rb / 1e6, wb / 1e6))
alg = 8 * N * D + 8 * K * D
print("attention total HBM-side traffic %.1f MB per call vs %.1f MB algorithmic (x%.2f)" % (tot / 1e6, alg / 1e6, tot / alg))
out.update(total_bytes=round(tot), algorithmic_bytes=alg, workload="%s N=%d D=%d h=%d K=%d, hl (split bf16) operands = 4 bytes per element" % (wl, N, D, h, K),
           kernel="+".join(names), fetch_bytes_per_unit=fr, write_bytes_per_unit=fw)
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)

#!/usr/bin/env python3
"""usage: python tools/pmc_traffic_summary.py <dir with fetch/ and write/ rocprofv3 pmc outputs> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
N, D, K = 32768, 768, 200
KNOWN = N * D * 4


def per_kernel(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in acc.items()}   # skip the first two (cold) launches


def pick(tab, sub):
    for k, v in tab.items():
        if sub in k:
            return v
    return None


fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
cal_r, cal_w = pick(fetch, "critic_kernel"), pick(write, "FillFunctor")
print("raw counter units (KiB-like): critic FETCH_SIZE %.1f (known read %d B) | fill WRITE_SIZE %.1f (known write %d B)"
      % (cal_r, KNOWN, cal_w, KNOWN))
fr, fw = KNOWN / cal_r, KNOWN / cal_w      # bytes per counter unit, calibrated on known streaming traffic
print("bytes per counter unit: FETCH %.1f (1024 x %.2f) | WRITE %.1f (1024 x %.2f)" % (fr, fr / 1024, fw, fw / 1024))
out = {}
for name in ("sparse_attn_mfma_kernel", "reduce_partials_kernel"):
    rb, wb = pick(fetch, name) * fr, pick(write, name) * fw
    out[name] = {"read_bytes": round(rb), "write_bytes": round(wb)}
    print("%-28s read %8.2f MB  write %8.2f MB per launch" % (name, rb / 1e6, wb / 1e6))
tot = sum(v["read_bytes"] + v["write_bytes"] for v in out.values())
alg = 2 * N * D * 2 + K * D * (2 + 4)   # Q, V bf16 + Kp bf16 in, O f32 out
print("attention total HBM-side traffic %.1f MB per launch vs %.1f MB algorithmic (x%.2f)" % (tot / 1e6, alg / 1e6, tot / alg))
out["total_bytes"] = tot
out["algorithmic_bytes"] = alg
out["workload"] = "cfgB N=32768 D=768 h=6 K=200 bf16 operands"
out["fetch_bytes_per_unit"], out["write_bytes_per_unit"] = fr, fw
out["kernel"] = "sparse_attn_mfma_kernel+reduce_partials_kernel"
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
# the fp32-class kernel (split-bf16 x 3): fp32 Q, V, Kp in, O out
if pick(fetch, "sparse_attn_x3_kernel") is not None:
    o3 = {}
    for name in ("sparse_attn_x3_kernel", "x3_reduce_kernel"):
        rb, wb = pick(fetch, name) * fr, pick(write, name) * fw
        o3[name] = {"read_bytes": round(rb), "write_bytes": round(wb)}
        print("%-28s read %8.2f MB  write %8.2f MB per launch" % (name, rb / 1e6, wb / 1e6))
    tot3 = sum(v["read_bytes"] + v["write_bytes"] for v in o3.values())
    alg3 = 2 * N * D * 4 + 2 * K * D * 4
    print("x3 attention total HBM-side traffic %.1f MB per launch vs %.1f MB algorithmic (x%.2f)" % (tot3 / 1e6, alg3 / 1e6, tot3 / alg3))
    o3.update(total_bytes=tot3, algorithmic_bytes=alg3, workload="cfgB N=32768 D=768 h=6 K=200 fp32 operands",
              kernel="sparse_attn_x3_kernel+x3_reduce_kernel", fetch_bytes_per_unit=fr, write_bytes_per_unit=fw)
    if len(sys.argv) > 3:
        json.dump(o3, open(sys.argv[3], "w"), indent=1)

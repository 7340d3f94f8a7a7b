#!/bin/bash
# MFMA-pipe utilisation of the fp32-class training step's kernels (one SQ counter pass, kernel-trace only, over a few bench steps),
# summarised per kernel like tools/pmc_vit.sh.  usage (GPU box, repo root):   bash tools/pmc_train.sh gpurun_out/pmc_train
set -u
OUT=$1; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $ROOT/$OUT/sq -o pmc -- python $ROOT/bench.py --mode train --precision fp32 --steps 4 --warmup 2 --headline-only --no-cpu-baseline --no-roofline > $ROOT/$OUT/sq.log 2>&1
cd $ROOT && python - <<PY
import csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open("$OUT/sq/pmc_counter_collection.csv")):
    acc[row["Kernel_Name"][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for row in csv.DictReader(open("$OUT/sq/pmc_kernel_trace.csv")):
    dur[row["Kernel_Name"][:70]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("%-72s %6s %9s %14s %12s" % ("kernel", "calls", "avg us", "MFMA insts", "MFMA busy %"))
for k, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    c = acc.get(k, {})
    mf = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) ; n = max(1, len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])))
    us = sum(d) / len(d)
    # busy cycles are summed over the 1024 SIMDs of the chip; the pipe of one SIMD is busy at most (duration x clock) cycles
    pct = 100.0 * (mf / n) / (1024 * us * 2.4e3) if us > 0 else 0.0
    if sum(d) > 300:
        print("%-72s %6d %9.1f %14.0f %11.1f%%" % (k, len(d), us, sum(c.get("SQ_INSTS_MFMA", [0])) / n, pct))
PY

#!/bin/bash
# per-kernel durations (rocprofv3 --kernel-trace --stats) of the attention micro-benchmark for each given library build
# usage: bash tools/ab_prof.sh <lib.so>...      (on the GPU box, from the repo root)
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  name=$(basename $L .so)
  SNUFFY_HIP_LIB=$(realpath $ROOT/$L) rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/abprof_$name -o p -- python $ROOT/tools/kbench.py attnB > /dev/null 2>&1
  echo "== $name"
  python - <<PY
import csv,glob
for f in glob.glob("$ROOT/gpurun_out/abprof_$name/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sparse_attn" in r["Name"] or "reduce_partials" in r["Name"]:
            print("   %-60s calls %4s avg %8.1f us  min %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done

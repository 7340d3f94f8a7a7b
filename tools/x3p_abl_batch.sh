#!/bin/bash
# development: a batch of timing ablations / knob settings of the pipelined attention kernel (config B shape unless X3P_SHAPE is set).
# usage (GPU box): bash tools/x3p_abl_batch.sh <kbw> "<defs 1>" "<defs 2>" ...     (results are wrong with X3P_ABL_* defines)
# Prints the unit time by HIP events (interquartile mean of 100 launches on cold rotating operands) and the main kernel's average
# duration under rocprofv3 (the figure to compare builds with: +-0.5 us).
# TRACE=1 adds the in-kernel s_memtime trace of one iteration (the stamps drain the LDS queue: read them for shape, not for time)
cd "$(dirname "$0")/.."
ROOT=$(pwd)
KBW=$1; shift
for D in "$@"; do
  touch snuffy_amd/csrc/sparse_attn_x3p_impl.h
  SNF_ATTN_DEV=1 SNF_EXTRA_DEFS="${TRACE:+X3P_TRACE }$D" python -c "from snuffy_amd.build import build_lib; build_lib()" > /dev/null 2>&1 || { echo "build failed: $D"; continue; }
  EV=$(python tools/x3p_dev.py ${X3P_SHAPE:-32768 200 6} --time --kbw=$KBW 2>&1 | grep "x3_hl" | sed 's/ -> .*//; s/.*median/median/')
  rm -rf /tmp/x3p_prof; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/x3p_prof -o p -- python $ROOT/tools/x3p_dev.py ${X3P_SHAPE:-32768 200 6} --time --kbw=$KBW > /dev/null 2>&1)
  KT=$(python - <<PY
import csv, glob
for f in glob.glob("/tmp/x3p_prof/**/p_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sparse_attn_x3p_kernel" in r["Name"] and int(r["Calls"]) > 50:
            print("kernel %.2f us (min %.1f, %s calls)" % (float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r["Calls"]))
PY
)
  echo "== kbw=$KBW defs: [$D] $KT | events: $EV"
  if [ -n "$TRACE" ]; then python tools/x3p_trace.py 100 $KBW 2>&1 | grep -A9 "^wave [03]$" | grep "^wave\|it  6\|fine" | head -8; fi
done

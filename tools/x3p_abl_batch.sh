#!/bin/bash
# development: a batch of timing ablations / knob settings of the pipelined attention kernel (config B shape unless X3P_SHAPE is set).
# usage (GPU box): bash tools/x3p_abl_batch.sh <kbw> "<defs 1>" "<defs 2>" ...     (results are wrong with X3P_ABL_* defines)
# TRACE=1 adds the in-kernel s_memtime trace of one iteration (the stamps drain the LDS queue: read them for shape, not for time)
cd "$(dirname "$0")/.."
KBW=$1; shift
for D in "$@"; do
  touch snuffy_amd/csrc/sparse_attn_x3p_impl.h
  SNF_ATTN_DEV=1 SNF_EXTRA_DEFS="${TRACE:+X3P_TRACE }$D" python -c "from snuffy_amd.build import build_lib; build_lib()" > /dev/null 2>&1 || { echo "build failed: $D"; continue; }
  echo "== kbw=$KBW defs: [$D] $(python tools/x3p_dev.py ${X3P_SHAPE:-32768 200 6} --time --kbw=$KBW 2>&1 | grep "x3_hl" | sed 's/ -> .*//; s/.*median/median/')"
  if [ -n "$TRACE" ]; then python tools/x3p_trace.py 100 $KBW 2>&1 | grep -A9 "^wave [03]$" | grep "^wave\|it  6\|fine" | head -8; fi
done

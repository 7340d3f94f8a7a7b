#!/bin/bash
# development: build the config-B variant of the pipelined attention kernel with ablation defines and time / trace it
# usage (on the GPU box): tools/x3p_abl.sh "<defs>" [key blocks per wave: 1 | 2] [trace-wg]
set -e
cd "$(dirname "$0")/.."
touch snuffy_amd/csrc/sparse_attn_x3p_impl.h
SNF_ATTN_DEV=1 SNF_EXTRA_DEFS="$1" python -c "from snuffy_amd.build import build_lib; build_lib()" 
echo "== defs: [$1] kbw=${2:-1}"
if [ -n "$3" ]; then
  python tools/x3p_trace.py $3 ${2:-1} 2>&1 | grep -v amdgpu.ids
else
  python tools/x3p_dev.py ${X3P_SHAPE:-32768 200 6} --time --kbw=${2:-1} 2>&1 | grep -v amdgpu.ids | grep "x3_hl"
fi

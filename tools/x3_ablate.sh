#!/bin/bash
# Timing ablations of sparse_attn_x3_kernel (numerically wrong on purpose; never shipped): what the in-kernel split of Q / V costs.  Builds side copies of the library under /tmp, runs tools/kbench.py x3B against each.
#   X3_ABL_NOSPLIT  split8 -> two cheap packs (keeps the data dependence and the LDS stores)
set -u
ROOT=$(pwd)
for V in ${VARIANTS:-BASE X3_ABL_NOSPLIT}; do
  mkdir -p /tmp/x3abl_$V
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -fno-honor-nans -D$V -c snuffy_amd/csrc/sparse_attn_x3.hip -o /tmp/x3abl_$V/x3.o || exit 1
  OBJS=$(ls snuffy_amd/build/*.o | grep -v sparse_attn_x3.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/x3abl_$V/libsnuffy_hip.so $OBJS /tmp/x3abl_$V/x3.o || exit 1
  echo "== $V"
  SNUFFY_HIP_LIB=/tmp/x3abl_$V/libsnuffy_hip.so python tools/kbench.py x3B 2>&1 | grep attn_x3
done

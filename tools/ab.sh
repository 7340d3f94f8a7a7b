#!/bin/bash
# A/B timing of two builds of libsnuffy_hip.so on the same box, alternating processes (box-to-box clocks differ by +-6 %).
# usage: bash tools/ab.sh <libA.so> <libB.so> [kbench target] [rounds]
A=$1; B=$2; WHAT=${3:-attnB}; N=${4:-3}
for i in $(seq $N); do
  for L in $A $B; do
    echo -n "$(basename $L): "; SNUFFY_HIP_LIB=$(realpath $L) python tools/kbench.py $WHAT 2>/dev/null | grep -v amdgpu.ids | tr '\n' ' '; echo
  done
done

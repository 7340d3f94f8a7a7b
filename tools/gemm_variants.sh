#!/bin/bash
# Timing ablations of the GEMM kernel: builds libsnuffy_hip variants with one part of gemm.hip compiled out
# (run here, they travel to the GPU box under snuffy_amd/build/variants/), then on the box:
#   for v in base nostore nomfma nostage nolds; do SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_$v.so python tools/gemm_bench.py cfgB; done
set -e
cd "$(dirname "$0")/.."
python -m snuffy_amd.build >/dev/null
mkdir -p snuffy_amd/build/variants
OBJS=$(ls snuffy_amd/build/*.o | grep -v gemm.o)
for v in base nostore nomfma nostage nolds "$@"; do
  D=""; [ "$v" != base ] && D="-DSNF_GEMM_$(echo $v | tr a-z A-Z)"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $D -c snuffy_amd/csrc/gemm.hip -o snuffy_amd/build/variants/gemm_$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o snuffy_amd/build/variants/lib_$v.so $OBJS snuffy_amd/build/variants/gemm_$v.o
done
ls -la snuffy_amd/build/variants/*.so

# A / B of dev builds against the shipped library, interleaved (same box, same clocks): bash tools/gemm_x3_ablate.sh  [VARIANTS="a b"]
for rep in 1 2; do
echo "== shipped"; timeout 120 python tools/gemm_x3_bench.py cfgB 2>&1 | grep one | sed 's/.*x3 one pass/one pass/'
for v in ${VARIANTS:-hl_dma_after_reads}; do
  echo "== $v"; SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_$v.so timeout 120 python tools/gemm_x3_bench.py cfgB 2>&1 | grep one | sed 's/.*x3 one pass/one pass/'
done
done

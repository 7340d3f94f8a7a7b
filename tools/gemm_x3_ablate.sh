echo "== full"; timeout 120 python tools/gemm_x3_bench.py cfgB 2>&1 | sed 's/.*x3 one pass/one pass/'
for v in ${VARIANTS:-hl_nt}; do
  echo "== $v"; SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_$v.so timeout 120 python tools/gemm_x3_bench.py cfgB 2>&1 | sed 's/.*x3 one pass/one pass/'
done

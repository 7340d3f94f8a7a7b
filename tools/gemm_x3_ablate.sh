echo "== full"; timeout 120 python tools/gemm_x3_bench.py cfgB 2>&1 | grep "Q|V" | sed 's/.*x3 one pass/one pass/'
for v in x3_nomfma x3_noread x3_nostage x3_nostore x3_stageonly x3_mfmaonly x3_nostore_nomfma; do
  echo "== $v"; SNUFFY_HIP_LIB=snuffy_amd/build/variants/lib_$v.so timeout 120 python tools/gemm_x3_bench.py cfgB 2>&1 | grep "Q|V" | sed 's/.*x3 one pass/one pass/'
done

#!/bin/bash
# SQ counter passes over one GEMM shape (kernel-trace only).  usage (GPU box, repo root): bash tools/pmc_gemm.sh <outdir> m n k [tile_n]
set -u
OUT=$1; shift
ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -o pmc -- python $ROOT/tools/gemm_one.py $ARGS > $ROOT/$OUT/$name.log 2>&1; }
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_UNALIGNED_STALL
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
run fetch FETCH_SIZE
cd $ROOT && python tools/pmc_summary.py $OUT gemm_bf16

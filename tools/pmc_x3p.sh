#!/bin/bash
# PMC passes over the pipelined attention kernel at config B (separate passes; kernel-trace only -- never with sys/hip tracing).
# usage (on the GPU box, from repo root): bash tools/pmc_x3p.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_x3p}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -o pmc -- python $ROOT/tools/x3p_dev.py 32768 200 6 --time > $ROOT/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
run sq3 SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $ROOT
python tools/pmc_summary.py $OUT x3p_kernel > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

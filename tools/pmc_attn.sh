#!/bin/bash
# PMC passes over the attention micro-benchmark (separate passes; kernel-trace only -- never with sys/hip tracing).
# usage (on the GPU box, from repo root): bash tools/pmc_attn.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/$name -o pmc -- python $ROOT/tools/kbench.py ${WHAT:-attnB} > $ROOT/$OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM
#run fetch FETCH_SIZE
#run write WRITE_SIZE
#run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
#run grbm GRBM_GUI_ACTIVE GRBM_COUNT

#!/bin/bash
# HBM-traffic counters of the attention kernels: one rocprofv3 --pmc pass per counter, kernel-trace only.
# usage (GPU box, repo root): bash tools/pmc_traffic.sh gpurun_out/traffic
OUT=${1:-gpurun_out/traffic}; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/fetch -o pmc -- python $ROOT/tools/pmc_traffic.py > $ROOT/$OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/write -o pmc -- python $ROOT/tools/pmc_traffic.py > $ROOT/$OUT/write.log 2>&1
cd $ROOT && python tools/pmc_traffic_summary.py $OUT $OUT/attn_traffic.json $OUT/attn_traffic_fp32.json

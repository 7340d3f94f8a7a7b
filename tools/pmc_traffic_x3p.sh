#!/bin/bash
# HBM-traffic counters of the pipelined fp32-class attention: one rocprofv3 --pmc pass per counter, kernel-trace only.
# usage (GPU box, repo root): bash tools/pmc_traffic_x3p.sh cfgB|cfgC gpurun_out/traffic_x3p_cfgB
WLN=${1:-cfgB}; OUT=${2:-gpurun_out/traffic_x3p_$WLN}; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/fetch -o pmc -- python $ROOT/tools/pmc_traffic_x3p.py $WLN > $ROOT/$OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/write -o pmc -- python $ROOT/tools/pmc_traffic_x3p.py $WLN > $ROOT/$OUT/write.log 2>&1
cd $ROOT && python tools/pmc_traffic_x3p_summary.py $OUT $WLN $OUT/attn_traffic_${WLN}_fp32.json

#!/bin/bash
# rocprofv3 kernel stats of any command.  usage: bash tools/prof_cmd.sh <outname> <command ...>
ROOT=$(pwd); NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
( cd $ROOT && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$NAME -o p -- "$@" > $ROOT/gpurun_out/prof_$NAME.log 2>&1 )
python - <<PY
import csv,glob
for f in glob.glob("$ROOT/gpurun_out/prof_$NAME/**/p_kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("== $NAME  total kernel time %.1f ms" % (tot/1e6))
    for r in rows[:22]:
        print("  %5.1f%% calls %5s avg %8.1f us  %s" % (100*float(r["TotalDurationNs"])/tot, r["Calls"], float(r["AverageNs"])/1e3, r["Name"][:110]))
PY

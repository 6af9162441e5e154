#!/bin/bash
# rocprofv3 kernel stats of a bench.py run: tools/prof_bench.sh <tag> <bench.py flags...>   -> gpurun_out/prof_<tag>/
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 bench.py "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:14]:
    print("%-72s %6s %12.1f us %6s %%" % (row["Name"][:72], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
PY

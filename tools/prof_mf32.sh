#!/bin/bash
# rocprofv3 kernel stats of the bench step: tools/prof_mf32.sh <tag> <slots> <precision 1|2>
export DBAZ_LIB=${DBAZ_LIB:-$PWD/dotsboxesaz_amd/libdbaz_hip_debug.so}   # the A/B tilings live in the debug build (python -m dotsboxesaz_amd.build --debug)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 bench.py --gpus 1 --steps 60 --warmup 10 --slots $2 --precision $3 --population fresh --no-tt --no-cpu-baseline --no-f32-side-run --games-leg 0 > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:12]:
    print(row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"])
PY

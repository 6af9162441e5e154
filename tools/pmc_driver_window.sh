#!/bin/bash
# PMC counters of the main k_tower launch over the TIMED WINDOW of the driver's own bench command (its last K launches):
#   bash tools/pmc_driver_window.sh [steps] [warmup]     -> gpurun_out/pmc_driver_window/summary.json (committed as profiles/r03_pmc_tower_driver_window.json)
# One counter group per rocprofv3 run, --kernel-trace only beside --pmc (MI355X_MICROARCH.md, HBM / rocprofv3 section).
K=${1:-20}; W=${2:-5}
repo=$PWD
OUT=$repo/gpurun_out/pmc_driver_window
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $repo
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --gpus 1 --steps $K --warmup $W --games-leg 0 --no-cpu-baseline --no-f32-side-run > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
  echo "pass $i ($grp) done"
done
python3 tools/pmc_summary.py $OUT $OUT/summary.json --last $K   # copy to profiles/r03_pmc_tower_driver_window.json
tail -n 1 $OUT/pass1.json > $OUT/bench_line_under_pmc.json
find $OUT -name "*.csv" -size +1M -delete

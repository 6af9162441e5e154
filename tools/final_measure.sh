#!/bin/bash
# Round-end measurement session on the GPU box (one gpurun call): the driver's command, its rocprofv3 kernel trace + stats, the
# PMC counters of its timed window (tied to the build by dbaz_build_info), the sustained 200-step figure and the other configs.
#   bash tools/final_measure.sh r03      -> gpurun_out/final_<tag>/...  (copy the summaries into profiles/)
TAG=${1:-r03}
repo=$PWD
OUT=$repo/gpurun_out/final_$TAG
rm -rf $OUT; mkdir -p $OUT
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench (driver command) done"
python bench.py --gpus 1 --steps 200 --warmup 10 --games-leg 0 --no-cpu-baseline --no-f32-side-run > $OUT/bench_200_steps.json 2>/dev/null
python bench.py --gpus 1 --steps 200 --warmup 10 --board 3 --slots 4096 --sims 100 --games-leg 0 --no-cpu-baseline --no-f32-side-run > $OUT/bench_config1_3x3_4096_100.json 2>/dev/null
python bench.py --gpus 1 --steps 200 --warmup 10 --board 9 --slots 2048 --sims 1600 --games-leg 0 --no-cpu-baseline --no-f32-side-run > $OUT/bench_config4_9x9_2048_1600.json 2>/dev/null
echo "other configs done"
bash tools/prof_bench.sh $TAG --gpus 1 --steps 20 --warmup 5 --games-leg 0 --no-cpu-baseline --no-f32-side-run > $OUT/prof_top.txt 2>&1
P=$repo/gpurun_out/prof_$TAG
python3 tools/trace_window.py $(find $P -name "*kernel_trace.csv" | head -1) $P/bench.json 20 $OUT/kernel_trace_timed_window.json >> $OUT/prof_top.txt 2>&1
cp $(find $P -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $P/bench.json $OUT/bench_under_rocprof.json
echo "rocprof trace done"
bash tools/pmc_driver_window.sh 20 5 > $OUT/pmc.log 2>&1
cp $repo/gpurun_out/pmc_driver_window/summary.json $OUT/pmc_tower_driver_window.json
echo "pmc done"
find $repo/gpurun_out/prof_$TAG $repo/gpurun_out/pmc_driver_window -name "*.csv" -size +1M -delete
cat $OUT/prof_top.txt

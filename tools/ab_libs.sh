#!/bin/bash
# Same-box A/B of two builds of the library on the real workloads (bench.py, games population):
#   bash tools/ab_libs.sh <lib A> <lib B> [steps]      -> gpurun_out/ab_libs.txt
# alternating A, B, A, B per configuration (6x6 configs[2], 9x9 share of configs[4], 3x3 configs[1])
A=$1; B=$2; K=${3:-200}
OUT=gpurun_out/ab_libs.txt; mkdir -p gpurun_out; : > $OUT
run() { # tag lib args...
  local tag=$1 lib=$2; shift 2
  DBAZ_LIB=$lib python bench.py --steps $K --warmup 10 --games-leg 0 --no-cpu-baseline --no-f32-side-run "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline')
if r: print('%-3s %-40s %10.0f exp/s  %.4f ms/step  tower %.4f ms  frac %.3f' % ('$tag', '$*', d['value'], d['ms_per_step'], r.get('tower_ms_per_step', 0), r['frac']))
else: print('%-3s %-40s %.2f games/s  %.0f exp/s' % ('$tag', '$*', d['value'], d['expansions_per_sec']))" >> $OUT
}
for i in 1 2; do
  run A $A; run B $B
done
for i in 1 2; do
  run A $A --board 9 --slots 2048 --sims 1600; run B $B --board 9 --slots 2048 --sims 1600
done
for i in 1 2; do
  run A $A --board 3 --slots 4096 --sims 100; run B $B --board 3 --slots 4096 --sims 100
done
for i in 1 2; do
  run A $A --full-games 1024 --slots 1024; run B $B --full-games 1024 --slots 1024
done
cat $OUT

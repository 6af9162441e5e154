#!/bin/bash
# A/B of the f16x3 tower tilings on whole rounds of workgroups (no tails, no cache hits, fresh trees: every slot evaluates)
#   precision 4: 16x16x32, one cout tile per wave, 4 samples per workgroup -> 4096 slots = 4 rounds
#   precision 2: 32x32x16, 5 samples               -> 3840 slots = 3 rounds
#   precision 3: 16x16x32, two cout tiles per wave, 5 samples -> 3840 slots = 3 rounds
export DBAZ_LIB=${DBAZ_LIB:-$PWD/dotsboxesaz_amd/libdbaz_hip_debug.so}   # the A/B tilings live in the debug build (python -m dotsboxesaz_amd.build --debug)
run() {
python bench.py --gpus 1 --steps 60 --warmup 10 --slots $1 --precision $2 $3 --population fresh --no-tt --no-cpu-baseline --no-f32-side-run --games-leg 0 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); r=d['roofline']; ev=r['flops_per_launch']/144.5e6
print('precision %d %-14s slots %d evals/step %.0f tower %.1f us  -> %.4f us/eval  frac %.3f' % ($2, '$3', $1, ev, 1e3*r['tower_ms_per_step'], 1e3*r['tower_ms_per_step']/ev, r['frac']))"
}
for i in 1 2; do
run 4096 4; run 3840 2; run 3840 3
done
run 4096 4 --zero-weights; run 3840 2 --zero-weights; run 3840 3 --zero-weights

#!/bin/bash
# A/B of the k_tower variants of the DEBUG build (python -m dotsboxesaz_amd.build --debug) on whole rounds of workgroups
# (3840 slots = 3 rounds of 1280 evaluations, no transposition table, fresh trees: every slot evaluates every step):
#   precision 3: the shipped two-cout-tile kernel (VAR 0)      5: + residual stream in f32 registers (VAR 1)
#   6: + XOR-swizzled column chunks (VAR 2)   7: both (VAR 3)  8: s_setprio 1 for waves 4-7 (VAR 8)   9: all three (VAR 11)
#   10: weight fragments through a two-slot LDS-DMA ring (VAR 4)   11: that + register residual (VAR 5)
# Usage on the GPU box: bash tools/ab_variants.sh "3 5 6 7 8 9" [rounds]     -> gpurun_out/ab_variants.txt
export DBAZ_LIB=$PWD/dotsboxesaz_amd/libdbaz_hip_debug.so
VARS=${1:-"3 5 6 7 8 9"}; ROUNDS=${2:-2}; export AB_VARS="$VARS"
OUT=gpurun_out/ab_variants.txt
mkdir -p gpurun_out; : > $OUT
python - <<'PY' >> $OUT 2>&1
# parity of every variant first: 1500 positions, 6x6 20x64, against torch fp32 (tolerance 1e-4; observed ~1e-6)
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import nn_ref
from dotsboxesaz_amd.engine import Engine
torch.manual_seed(0)
m = nn_ref.ResNetZeroRef(6, 6, 64, 20); nn_ref.randomize_bn(m, 3)
rng = np.random.RandomState(5)
X = rng.randint(0, 2, size=(1500, 3, 7, 7)).astype(np.float32); X[:, 2] = rng.randint(-1, 37, size=(1500, 1, 1))
pr, vr = nn_ref.predict_sync(m, X)
base = None
for prec in [int(x) for x in os.environ.get('AB_VARS', '3 5 6 7 8 9 10 11').split()]:
    e = Engine(6, 6, 2048, evaluator="resnet", nn_precision=prec)
    e.load_state_dict(m.state_dict(), "resnet", 64, 20, 16, 8)
    p, v = e.predict(X); e.close()
    if base is None: base = (p, v)
    print("parity precision %d: max |d(p,v)| vs torch fp32 %.2e, vs the shipped kernel %.2e" % (prec, max(np.abs(p - pr).max(), np.abs(v - vr).max()), max(np.abs(p - base[0]).max(), np.abs(v - base[1]).max())))
PY
run() {
python bench.py --gpus 1 --steps 60 --warmup 10 --slots 3840 --precision $1 $2 --population fresh --no-tt --no-cpu-baseline --no-f32-side-run --games-leg 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; ev=r['flops_per_launch']/144.5e6
print('precision $1 %-14s evals/step %.0f tower %.1f us  -> %.4f us/eval  frac %.3f' % ('$2', ev, 1e3*r['tower_ms_per_step'], 1e3*r['tower_ms_per_step']/ev, r['frac']))" >> $OUT
}
for i in $(seq $ROUNDS); do for v in $VARS; do run $v; done; done
for v in $VARS; do run $v --zero-weights; done
cat $OUT

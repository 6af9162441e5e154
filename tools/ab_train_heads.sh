#!/bin/bash
# Same-box A/B of the training step: whole network on csrc/train.hip (A) vs stem and heads on torch (B), alternating
#   bash tools/ab_train_heads.sh [rounds]     -> gpurun_out/ab_train_heads.txt
R=${1:-3}
OUT=gpurun_out/ab_train_heads.txt; mkdir -p gpurun_out; : > $OUT
run() {
  python tools/train_step_time.py 4096 --steps 40 $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-2s %.3f ms/step  (data %.2f ms, host fwd/bwd/opt/read %s)  %s' % ('$1', d['ms_per_step'], d['ms_data_per_step'], d['host_ms_fwd_bwd_opt_read'], d['what'][38:]))" >> $OUT
}
for i in $(seq $R); do run A ""; run B --torch-heads; done
cat $OUT

#!/bin/bash
# Same-box A/B of two builds of the library on the training step (tools/train_step_time.py, batch 4096, 40 steps), alternating:
#   bash tools/ab_train_step.sh <lib A> <lib B> [rounds]     -> gpurun_out/ab_train_step.txt
A=$1; B=$2; R=${3:-3}
OUT=gpurun_out/ab_train_step.txt; mkdir -p gpurun_out; : > $OUT
run() {
  DBAZ_LIB=$2 python tools/train_step_time.py 4096 --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-2s %.3f ms/step  (data %.2f ms, host fwd/bwd/opt/read %s)' % ('$1', d['ms_per_step'], d['ms_data_per_step'], d['host_ms_fwd_bwd_opt_read']))" >> $OUT
}
for i in $(seq $R); do run A $A; run B $B; done
cat $OUT

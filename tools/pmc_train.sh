#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, --kernel-trace only beside --pmc) over a short training-step run.
#   tools/pmc_train.sh <tag> [train_step_time.py arguments]   -> gpurun_out/pmc_<tag>/pass*/...
# then: python tools/pmc_summary.py gpurun_out/pmc_<tag> profiles/<name>.json
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pass$i -o p -- python3 tools/train_step_time.py "$@" > $OUT/pass$i.json 2> $OUT/pass$i.err || echo "pass $i failed"
  rm -f $OUT/pass$i/*/*kernel_trace.csv $OUT/pass$i/*kernel_trace.csv
  echo "pass $i ($grp) done"
done

#!/bin/bash
# rocprofv3 kernel statistics of the training step (tools/train_step_time.py); run on the GPU box from the repo root:
#   bash tools/prof_train.sh <tag> [train_step_time.py arguments]
tag=$1; shift
out=$PWD/gpurun_out/prof_train_$tag
mkdir -p $out
repo=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $repo/tools/train_step_time.py "$@" > $out/step.json 2> $out/err.txt
cd $repo
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/p_kernel_stats.csv")))
for r in rows[:22]:
    print("%-60s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
python3 - <<PY
# k_conv_t launches in start order alternate 40 forward / 40 input-gradient launches per step (20 blocks)
import csv
rows = [r for r in csv.DictReader(open("$out/p_kernel_trace.csv")) if r["Kernel_Name"].startswith("k_conv_t")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
L = 40
fwd = [x for i, x in enumerate(d) if (i % (2 * L)) < L]
bwd = [x for i, x in enumerate(d) if (i % (2 * L)) >= L]
if fwd and bwd:
    print("k_conv_t forward launches: avg %.1f us; input-gradient launches: avg %.1f us (with + g: %.1f, without: %.1f)"
          % (sum(fwd) / len(fwd), sum(bwd) / len(bwd), sum(bwd[1::2]) / len(bwd[1::2]), sum(bwd[0::2]) / len(bwd[0::2])))
PY
python3 - <<PY
# the heads' GEMM launches in start order: per step fc forward, fc weight gradient, fc input gradient, head conv weight gradient(, stem)
import csv
rows = [r for r in csv.DictReader(open("$out/p_kernel_trace.csv")) if r["Kernel_Name"].startswith("k_gemm_f32")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
steps = len([r for r in csv.DictReader(open("$out/p_kernel_trace.csv")) if r["Kernel_Name"].startswith("k_head_out_bwd")])
if steps and d and len(d) % steps == 0:
    per = len(d) // steps
    print("k_gemm_f32 per step, in launch order (us):", ["%.1f" % (sum(d[i::per]) / steps) for i in range(per)])
PY
rm -f $out/p_kernel_trace.csv
cat $out/step.json

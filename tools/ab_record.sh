#!/bin/bash
# A/B RECORD of the k_tower variants (debug build): per variant the PMC counters of the main launch on whole rounds
# (3840 slots = 3 rounds, fresh trees, no transposition table) -- traffic, MFMA busy fraction, LDS cycles and conflicts, and the
# clock the chip held under the counters (GRBM_GUI_ACTIVE / 8 / kernel duration).  Timing itself: tools/ab_variants.sh (un-profiled).
#   bash tools/ab_record.sh "3 5 6 10 1"     -> gpurun_out/ab_record/variant_<p>.json  (copy into profiles/r03_ab_*)
export DBAZ_LIB=$PWD/dotsboxesaz_amd/libdbaz_hip_debug.so
VARS=${1:-"3 5 6 10 1"}
repo=$PWD
OUT=$repo/gpurun_out/ab_record
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $repo
for v in $VARS; do
  i=0; mkdir -p $OUT/v$v
  for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/v$v/pass$i -o p -- python3 bench.py --gpus 1 --steps 30 --warmup 10 --slots 3840 --precision $v --population fresh --no-tt --no-cpu-baseline --no-f32-side-run --games-leg 0 > $OUT/v$v/pass$i.json 2> $OUT/v$v/pass$i.err || echo "variant $v pass $i failed"
  done
  python3 tools/pmc_summary.py $OUT/v$v $OUT/variant_$v.json --last 30 > /dev/null
  python3 - <<PY
import json
d = json.load(open("$OUT/variant_$v.json"))
k = max((n for n in d["kernels"] if n.startswith("k_tower")), key=lambda n: d["kernels"][n]["mean_duration_ns_under_pmc"])
t = d["kernels"][k]
dur = t["mean_duration_ns_under_pmc"]
clk = t.get("GRBM_GUI_ACTIVE", {}).get("mean", 0) / 8.0 / dur * 1e3
print("variant %-3s %s: %.1f us under PMC, traffic %.1f MB, MFMA busy %.3f, LDS active %.0f M / conflict %.0f M cycles, clock %.0f MHz"
      % ("$v", k.split("<")[1].split(">")[0], dur / 1e3, d.get("traffic_bytes_per_launch", 0) / 1e6, d.get("mfma_busy_frac", 0),
         t.get("SQ_LDS_IDX_ACTIVE", {}).get("mean", 0) / 1e6, t.get("SQ_LDS_BANK_CONFLICT", {}).get("mean", 0) / 1e6, clk))
PY
  find $OUT/v$v -name "*.csv" -size +200k -delete
done 2>&1 | tee $OUT/summary.txt

#!/bin/bash
# Diagnostic: builds a STAMPED copy of the library (never the shipped one) and prints where a
# k_select wave spends its cycles (root loads / descent / leaf creation + features).
# Usage on the GPU box: bash tools/stamp_select.sh
set -e
cd "$(dirname "$0")/.."
D=gpurun_out/stamp_build
mkdir -p $D
for f in tree engine nn replay; do
  extra=""; [ $f = tree ] && extra="-ffp-contract=off"; [ $f = replay ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DDBAZ_STAMP $extra -c dotsboxesaz_amd/csrc/$f.hip -o $D/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libdbaz_hip.so $D/tree.o $D/engine.o $D/nn.o $D/replay.o
python - > $D/select_stamps.txt <<'PY'
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from dotsboxesaz_amd import _lib
_lib.LIB_PATH = "gpurun_out/stamp_build/libdbaz_hip.so"
from dotsboxesaz_amd.engine import Engine
e = Engine(6, 6, 8192, mcts_num_read=800, noise=(0.8, 0.25), evaluator="uniform", seed=1)
span = int(0.7 * e.E)
e.selfplay_fastforward((np.arange(8192) * 37) % span)
e.selfplay_start(1 << 40, 0)
e.step(400)
e.sync()
e.close()
PY
python - <<'PY'
import re, numpy as np
rows = [tuple(map(int, re.findall(r"\d+", l))) for l in open("gpurun_out/stamp_build/select_stamps.txt") if l.startswith("SEL")]
a = np.array(rows[len(rows) // 2:], dtype=np.float64)  # second half: trees have grown
print("samples", len(a), "mean depth %.2f" % a[:, 1].mean())
for i, n in enumerate(["root loads", "descent", "leaf (init_node + features)"]):
    print("%-30s mean %8.0f cycles  (%.1f us @2.4GHz)" % (n, a[:, 2 + i].mean(), a[:, 2 + i].mean() / 2400))
print("descent per level: %.0f cycles" % (a[:, 3].sum() / np.maximum(1, a[:, 1]).sum()))
if a.shape[1] > 5:
    print("of which waiting for the next node's meta+rows (one dependent round trip per level): mean %.0f cycles per wave, %.0f per level descended"
          % (a[:, 5].mean(), a[:, 5].sum() / np.maximum(1, a[:, 1] - 1).sum()))
if a.shape[1] > 8:
    lv = np.maximum(1, a[:, 1]).sum()
    print("per level: pb_c/sqrt table lookup %.0f, UCB scan %.0f, argmax %.0f cycles" % (a[:, 6].sum() / lv, a[:, 7].sum() / lv, a[:, 8].sum() / lv))
PY

#!/bin/bash
# Diagnostic: builds a STAMPED copy of the library (never the shipped one) and prints where a
# k_tower wave spends its cycles.  Usage on the GPU box: bash tools/stamp_build_run.sh
set -e
cd "$(dirname "$0")/.."
# the stamped library is built in the build container: tools/build_stamp.sh -> build/stamp/libdbaz_hip.so
python - <<'PY'
import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, ".")
from dotsboxesaz_amd import _lib
_lib.LIB_PATH = "build/stamp/libdbaz_hip.so"
from dotsboxesaz_amd.engine import Engine
from dotsboxesaz_amd import nn as dnn
import os
e = Engine(6, 6, 8192, evaluator="resnet", nn_precision=2 if os.environ.get("DBAZ_MF32") else (4 if os.environ.get("DBAZ_CLASSIC") else 1))
torch.manual_seed(0)
m = dnn.ResNetZero(dnn.resnet_params(6, 6))
e.load_state_dict(m.state_dict(), "resnet", **m.shape)
X = np.random.RandomState(0).randint(0, 2, size=(8192, 3, 7, 7)).astype(np.float32)
for _ in range(60):   # ~0.2 s of back-to-back launches so that the clock settles
    e.predict(X)
n_wg = 2048
out = np.zeros((n_wg, 8, 10), np.uint64)
L = _lib.load()
L.dbaz_debug_read_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = L.dbaz_debug_read_stamps(e.h, out.ctypes.data, n_wg)
o = out.astype(np.float64)
# only workgroups of the MAIN launch that ran: rows of idle entries are zero, and the first `cus` rows were overwritten by a
# tail launch (smaller workgroups) whenever the batch left a tail
ran = o[:, 0, 4] > 0
ran[:256] = False
o = o[ran]
tot = o[..., 4]
print("rc", rc, "waves", (tot > 0).sum())
names = ["prologue(loads)", "main loop", "epilogue", "barrier wait", "layers total"]
for i, nme in enumerate(names):
    print("%-18s mean %10.0f cycles/wave   %5.1f %% of layers total" % (nme, o[..., i].mean(), 100 * o[..., i].mean() / tot.mean()))
import os
if os.environ.get("DBAZ_MF32"):
    print("32x32x16 tiling, 5 samples per workgroup; per layer: total %.0f, main %.0f (MFMA floor per wave: 36 steps x 6 MFMAs x 32 = %d; per SIMD twice that)" % (tot.mean() / 40, o[..., 1].mean() / 40, 36 * 6 * 32))
elif os.environ.get("DBAZ_CLASSIC"):
    print("16x16x32, one cout tile per wave, 4 samples per workgroup; per layer: total %.0f, main %.0f (MFMA floor 7 tiles: %d, 6 tiles: %d; per SIMD their sum)" % (tot.mean() / 40, o[..., 1].mean() / 40, 18 * 21 * 16, 18 * 18 * 16))
else:
    print("16x16x32, two cout tiles per wave, 5 samples per workgroup; per layer: total %.0f, main %.0f (MFMA floor per wave: 18 steps x 24 MFMAs x 16 = %d; per SIMD twice that)" % (tot.mean() / 40, o[..., 1].mean() / 40, 18 * 24 * 16))
for w in range(8):
    print("wave", w, ["%.0f" % (o[:, w, i].mean() / 40) for i in range(5)])
whole = o[..., 7].mean()
print("whole workgroup %.0f cycles: conv0 phase %.0f (%.1f %%), 40 layers %.0f (%.1f %%), head convs + output %.0f (%.1f %%)"
      % (whole, o[..., 5].mean(), 100 * o[..., 5].mean() / whole, tot.mean(), 100 * tot.mean() / whole, o[..., 6].mean(),
         100 * o[..., 6].mean() / whole))
rt = o[..., 8]
ok = rt > 0
print("in-kernel clock (shader cycles / 100 MHz realtime ticks), median over workgroups: %.0f MHz" % (np.median(o[..., 7][ok] / rt[ok]) * 100))
PY

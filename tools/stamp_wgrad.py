"""Diagnostic (GPU box): where a k_wgrad_h3 workgroup spends its time.  Needs the stamped library (tools/build_stamp.sh):
DBAZ_LIB=$PWD/build/stamp/libdbaz_hip.so python tools/stamp_wgrad.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotsboxesaz_amd import _lib, nn as dnn, train_tower  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
dev = torch.device("cuda:0")
m = dnn.ResNetZero(dnn.resnet_params(6, 6)).to(dev)
m.train(True)
x = torch.randn(n, 64, 7, 7, device=dev, requires_grad=True)
for _ in range(3):
    y = train_tower.resblocks_forward(m, x)
    y.backward(torch.randn_like(y) * 1e-3)
torch.cuda.synchronize()
L = _lib.load()
t = [tr for per in train_tower._trainers.values() for tr in per.values()][0]
n_wg = 256
out = np.zeros((n_wg, 8, 8), np.uint64)
L.dbaz_debug_trainer_wgrad_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = L.dbaz_debug_trainer_wgrad_stamps(t.h, out.ctypes.data, n_wg)
o = out.astype(np.float64)
names = ["prologue (tables, zeroed images, first prefetch)", "commit (convert + LDS writes, 2 barriers) x chunks", "K loop x chunks", "partial store"]
tot = o[..., 4]
print("rc", rc, "cycles per workgroup (mean over waves): %.0f" % tot.mean())
for i, nm in enumerate(names):
    print("  %-52s %8.0f cycles  %5.1f %%   (min %.0f  max %.0f)" % (nm, o[..., i].mean(), 100 * o[..., i].mean() / tot.mean(), o[..., i].min(), o[..., i].max()))
rt0 = out[:, 0, 5].astype(np.int64)
rt1 = out[:, :, 6].astype(np.int64).max(1)
print("launch span %.1f us; workgroup duration mean %.1f us; clock %.0f MHz; per chunk (4 per workgroup): commit %.0f, K loop %.0f cycles (MFMA issue per wave: 7 K steps x 54 x 16 = 6048)"
      % ((rt1.max() - rt0.min()) / 100.0, ((rt1 - rt0) / 100.0).mean(), np.median(tot.mean(1) / ((rt1 - rt0) / 100.0)), o[..., 1].mean() / 4, o[..., 2].mean() / 4))

"""Diagnostic (GPU box): where a k_conv_t workgroup spends its time.  Needs the stamped library (tools/build_stamp.sh, built
in the build container): DBAZ_LIB=$PWD/build/stamp/libdbaz_hip.so python tools/stamp_conv_t.py [batch]"""
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
for _ in range(5):
    y = train_tower.resblocks_forward(m, x)
torch.cuda.synchronize()
L = _lib.load()
t = [tr for per in train_tower._trainers.values() for tr in per.values()][0]
S = 256 // 49
n_wg = (n + S - 1) // S
out = np.zeros((n_wg, 8, 8), np.uint64)
L.dbaz_debug_trainer_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rc = L.dbaz_debug_trainer_stamps(t.h, out.ctypes.data, n_wg)
o = out.astype(np.float64)
names = ["load+convert", "barrier 1", "MFMA loop", "epilogue->LDS (2 barriers)", "store"]
tot = o[..., :5].sum(-1)
print("rc", rc, "workgroups", n_wg, "cycles per workgroup (mean over waves): %.0f" % tot.mean())
for i, nm in enumerate(names):
    print("  %-28s %8.0f cycles  %5.1f %%   (min %.0f  max %.0f)" % (nm, o[..., i].mean(), 100 * o[..., i].mean() / tot.mean(),
                                                                   o[..., i].min(), o[..., i].max()))
rt0 = out[:, 0, 5].astype(np.int64)
rt1 = out[:, :, 6].astype(np.int64).max(1)
t0 = rt0.min()
start = (rt0 - t0) / 100.0   # us (100 MHz)
end = (rt1 - t0) / 100.0
print("launch span %.1f us; workgroup duration mean %.1f us (min %.1f max %.1f); clock %.0f MHz"
      % (end.max(), (end - start).mean(), (end - start).min(), (end - start).max(),
         np.median(tot.mean(1) / ((rt1 - rt0) / 100.0))))
hist, edges = np.histogram(start, bins=12)
print("start-time histogram (us):", [(round(float(e), 1), int(h)) for e, h in zip(edges[:-1], hist)])
ident = out[:, 0, 7]
xcc = (ident >> np.uint64(32)).astype(np.int64) & 0xF
hw = (ident & np.uint64(0xFFFFFFFF)).astype(np.int64)
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
sh = (hw >> 12) & 0x1
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("distinct CUs seen:", len(set(key.tolist())), " workgroups per CU: min %d max %d" % (np.bincount(key).min() if False else min(np.bincount(key)[np.bincount(key) > 0]), np.bincount(key).max()))
# timeline of one CU
k0 = key[0]
idx = np.where(key == k0)[0]
print("timeline of the CU of workgroup 0 (start, end us; phases in us at the measured clock):")
for i in idx[np.argsort(start[idx])]:
    clk = tot[i].mean() / ((rt1[i] - rt0[i]) / 100.0)
    print("   wg %4d  %6.1f -> %6.1f   " % (i, start[i], end[i]) + "  ".join("%s %.1f" % (nm.split()[0], o[i, :, j].mean() / clk) for j, nm in enumerate(names)))

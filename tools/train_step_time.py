#!/usr/bin/env python3
"""Times the optimizer step of the generation loop fed by the HIP batch kernel: 6x6, ResNetZero 20x64, batch 4096, SGD
momentum.  Residual tower forward/backward on csrc/train.hip (default) or on torch/MIOpen (--torch); heads, loss and
SGD are torch either way.  Prints one JSON line.

    python tools/train_step_time.py [batch] [--torch] [--steps K]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotsboxesaz_amd import nn as dnn  # noqa: E402
from dotsboxesaz_amd import train as T  # noqa: E402
from dotsboxesaz_amd.engine import Engine  # noqa: E402
from dotsboxesaz_amd.self_play import _DevBuf  # noqa: E402
from dotsboxesaz_amd.train_data import ReplayStore, SymmetriesGenerator  # noqa: E402

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(_pos[0]) if _pos else 4096
HIP = "--torch" not in sys.argv
K = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
e = Engine(6, 6, 256, mcts_num_read=12, noise=(0.8, 0.25), evaluator="uniform", seed=1)
e.selfplay_start(512, 0)
e.run()
ptr, n, rb = e.replay_rows_dev()
rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=torch.device("cuda", 0)).view(n, rb).clone()
store = ReplayStore(e)
store.add_generation(0, rows)
ds = store.dataset(train=True, pos_average=True)
params = dnn.resnet_params(6, 6, 64, 20)
torch.manual_seed(0)
model = dnn.ResNetZero(params).cuda()
model.train(True)
opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
crit = T.AlphaZeroLoss()
sym = SymmetriesGenerator(e)


def epoch(steps):
    done, t_data = 0, 0.0
    while done < steps:
        it = iter(ds.loader(B, True, True, sym))
        while done < steps:
            t0 = time.perf_counter()
            try:
                boards, pi, z = next(it)
            except StopIteration:
                break
            t_data += time.perf_counter() - t0
            p, v = T.training_forward(model, boards, hip_tower=HIP)
            loss, _ = crit(p, v, pi, z)
            loss.backward()
            opt.step()
            opt.zero_grad()
            done += 1
    torch.cuda.synchronize()
    return t_data


epoch(3)
t0 = time.perf_counter()
td = epoch(K)
dt = time.perf_counter() - t0
print(json.dumps({"what": "training step fed by k_make_batch; residual tower on %s, heads/loss/SGD on torch" % ("csrc/train.hip" if HIP else "torch (MIOpen)"), "board": "6x6", "net": "ResNetZero 20x64",
                  "batch": B, "dataset_rows": len(ds), "ms_per_step": 1e3 * dt / K, "ms_data_per_step": 1e3 * td / K,
                  "samples_per_sec": B * K / dt}))

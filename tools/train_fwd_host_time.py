#!/usr/bin/env python3
"""Diagnostic: host time of the pieces of training_forward in the optimizer-step loop (one full sync per step, like
tools/train_step_time.py), with torch's SGD and with HipSGD."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dotsboxesaz_amd import nn as dnn, train as T, train_tower

torch.manual_seed(0)
model = dnn.ResNetZero(dnn.resnet_params(6, 6, 64, 20)).cuda()
model.train(True)
B = 4096
pi = torch.softmax(torch.randn(B, 98), 1).cuda()
z = (torch.randint(0, 3, (B, 1)).float() - 1).cuda()
# batches from the engine's replay dataset, as tools/train_step_time.py takes them (argument "loader")
it_batches = None
if "loader" in sys.argv:
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd.self_play import _DevBuf
    from dotsboxesaz_amd.train_data import ReplayStore, SymmetriesGenerator
    e = Engine(6, 6, 256, mcts_num_read=12, noise=(0.8, 0.25), evaluator="uniform", seed=1)
    e.selfplay_start(512, 0)
    e.run()
    ptr, n, rb = e.replay_rows_dev()
    rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=torch.device("cuda", 0)).view(n, rb).clone()
    store = ReplayStore(e)
    store.add_generation(0, rows)
    ds = store.dataset(train=True, pos_average=True)
    sym = SymmetriesGenerator(e)

    def gen():
        while True:
            for b in ds.loader(B, True, True, sym):
                yield b
    it_batches = gen()
r, ph, vh = model.resnet, model.policy_head, model.value_head
bn = train_tower.batch_norm_train
for name, cls in (("torch", torch.optim.SGD), ("hip", T.HipSGD), ("torch", torch.optim.SGD), ("hip", T.HipSGD)):
    opt = cls(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    acc = [0.0] * 8
    for it in range(16):
        if it_batches is not None:
            boards, pi, z = next(it_batches)
        else:
            boards = torch.randn(B, 3, 7, 7, device="cuda")
        t = [time.perf_counter()]
        x = bn(model.bn_input, boards); t.append(time.perf_counter())
        x = r.conv0(x); t.append(time.perf_counter())
        x = bn(r.bn0, x, relu=True); t.append(time.perf_counter())
        x = train_tower.resblocks_forward(model, x); t.append(time.perf_counter())
        p = bn(ph.bn0, ph.conv0(x), relu=True)
        v = bn(vh.bn0, vh.conv0(x), relu=True); t.append(time.perf_counter())
        p = F.log_softmax(ph.fc(p.view(p.size(0), -1)), dim=1)
        v = F.relu(vh.fc0(v.view(v.size(0), -1)))
        v = torch.tanh(vh.fc1(v)); t.append(time.perf_counter())
        loss, parts = T.AlphaZeroLoss.tensors(p, v, pi, z)
        loss.backward(); t.append(time.perf_counter())
        opt.step(); opt.zero_grad(); t.append(time.perf_counter())
        parts[0].item()
        if it >= 4:
            for k in range(8):
                acc[k] += (t[k + 1] - t[k]) * 1e3 / 12
    print("%-5s host ms: bn_input %.2f conv0 %.2f bn0 %.2f tower %.2f head convs+bn %.2f head fcs %.2f loss+bwd %.2f opt %.2f"
          % (name, *acc))

#!/usr/bin/env python3
"""Diagnostic: host and device time of the phases of one optimizer step (forward, loss, backward, optimizer) for torch's SGD
and HipSGD.  python tools/train_phase_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dotsboxesaz_amd import nn as dnn, train as T

torch.manual_seed(0)
model = dnn.ResNetZero(dnn.resnet_params(6, 6, 64, 20)).cuda()
model.train(True)
B = 4096
boards = torch.randn(B, 3, 7, 7).cuda()
pi = torch.softmax(torch.randn(B, 98), 1).cuda()
z = (torch.randint(0, 3, (B, 1)).float() - 1).cuda()
class HipSuper(T.HipSGD):
    def step(self, closure=None):
        return torch.optim.SGD.step(self)


class HipTemps(T.HipSGD):
    """HipSGD + the temporaries torch's multi-tensor step allocates and frees (grads + wd * params)"""
    def step(self, closure=None):
        ps = [p for p in self.param_groups[0]["params"] if p.grad is not None]
        tmp = [torch.empty_like(p) for p in ps]
        r = super().step()
        del tmp
        return r


for name, cls in (("torch", torch.optim.SGD), ("hip", T.HipSGD), ("super", HipSuper), ("temps", HipTemps), ("torch", torch.optim.SGD), ("hip", T.HipSGD)):
    opt = cls(model.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    host = [0.0] * 4
    dev = [0.0] * 4
    for it in range(14):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        t = [time.perf_counter()]
        ev[0].record()
        p, v = T.training_forward(model, boards)
        ev[1].record(); t.append(time.perf_counter())
        loss, parts = T.AlphaZeroLoss.tensors(p, v, pi, z)
        ev[2].record(); t.append(time.perf_counter())
        loss.backward()
        ev[3].record(); t.append(time.perf_counter())
        opt.step()
        opt.zero_grad()
        ev[4].record(); t.append(time.perf_counter())
        torch.cuda.synchronize()
        if it >= 4:
            for k in range(4):
                host[k] += (t[k + 1] - t[k]) * 1e3 / 10
                dev[k] += ev[k].elapsed_time(ev[k + 1]) / 10
    print("%-5s host ms: fwd %.2f loss %.2f bwd %.2f opt %.2f | device ms: fwd %.2f loss %.2f bwd %.2f opt %.2f | sum device %.2f"
          % (name, *host, *dev, sum(dev)))

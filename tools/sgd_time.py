import sys, time, torch
sys.path.insert(0, ".")
from dotsboxesaz_amd import nn as dnn, train as T
m = dnn.ResNetZero(dnn.resnet_params(6, 6, 64, 20)).cuda()
ps = list(m.parameters())
for name, cls in (("torch", torch.optim.SGD), ("hip", T.HipSGD)):
    opt = cls(ps, lr=1e-2, momentum=0.9, weight_decay=1e-4)
    for p in ps: p.grad = torch.randn_like(p)
    for _ in range(3): opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): opt.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, "host per step %.3f ms, incl. device drain %.3f ms" % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))

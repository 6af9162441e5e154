"""The whole network of the optimizer step on HIP (csrc/train_net.hip + csrc/train.hip, dbaz_trainer_net_forward / _backward: bn_input, conv0 + bn0, the
residual tower, both heads) against torch autograd.

Ground truth = `train.training_forward(..., hip_tower=False)` -- the reference's operation order (nn.py:108-122) composed of torch
modules -- evaluated in FLOAT64 on the CPU, followed by the reference's AlphaZeroLoss (nn.py:131-138).  The HIP path works in f32
(f16x3 MFMA products in the tower, exact f32 elsewhere, f64 statistics) and must be as close to the float64 truth as torch's own
float32 evaluation is, within a small factor, for the outputs, EVERY parameter gradient and every running statistic."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_model(rows, cols, blocks, seed):
    from dotsboxesaz_amd import nn as dnn
    torch.manual_seed(seed)
    m = dnn.ResNetZero(dnn.resnet_params(rows, cols, 64, blocks))
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            c = mod.num_features
            mod.weight.data = torch.rand(c, generator=g) + 0.5
            mod.bias.data = torch.randn(c, generator=g) * 0.2
            mod.running_mean.data = torch.randn(c, generator=g) * 0.1
            mod.running_var.data = torch.rand(c, generator=g) + 0.5
    return m


def batch(rows, cols, n, A, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, 3, rows + 1, cols + 1, generator=g) < 0.4).float()
    pi = torch.softmax(torch.randn(n, A, generator=g) * 2.0, dim=1)
    z = (torch.rand(n, 1, generator=g) < 0.5).float() * 2.0 - 1.0
    return x, pi, z


def run(model, x, pi, z, dtype, device, **kw):
    from dotsboxesaz_amd import train as T
    m = copy.deepcopy(model).to(dtype).to(device)
    m.train(True)
    p, v = T.training_forward(m, x.to(dtype).to(device), **kw)
    loss, _ = T.AlphaZeroLoss.tensors(p, v, pi.to(dtype).to(device), z.to(dtype).to(device))
    loss.backward()
    if device != "cpu":
        torch.cuda.synchronize()
    grads = {k: q.grad.double().cpu() for k, q in m.named_parameters()}
    stats = {k: s.double().cpu() for k, s in m.state_dict().items() if "running" in k}
    nbt = sorted(set(int(s) for k, s in m.state_dict().items() if "num_batches_tracked" in k))
    return p.detach().double().cpu(), v.detach().double().cpu(), float(loss.detach()), grads, stats, nbt


def rel(a, ref):
    return float((a - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


@pytest.mark.parametrize("board,nb,n", [((6, 6), 2, 37), ((3, 3), 1, 70), ((9, 9), 1, 5), ((2, 3), 1, 9), ((6, 6), 3, 256), ((6, 6), 1, 1)])
def test_network_forward_backward_vs_float64(board, nb, n):
    from dotsboxesaz_amd import train_tower
    rows, cols = board
    model = make_model(rows, cols, nb, 11 * nb + n)
    A = model.policy_head.fc.out_features
    x, pi, z = batch(rows, cols, n, A, 5 + n)
    assert train_tower.net_supported(copy.deepcopy(model).cuda().train(True), x.cuda())
    p64, v64, l64, g64, s64, _ = run(model, x, pi, z, torch.float64, "cpu", hip_tower=False)
    p32, v32, l32, g32, s32, _ = run(model, x, pi, z, torch.float32, "cpu", hip_tower=False)
    ph, vh, lh, gh, sh, nbt = run(model, x, pi, z, torch.float32, "cuda", hip_tower=True, hip_heads=True)
    assert nbt == [1]
    assert ph.shape == (n, A) and vh.shape == (n, 1)

    def check(name, hip, t32, t64, floor):
        e_hip, e_t32 = rel(hip, t64), rel(t32, t64)
        assert e_hip <= max(4 * e_t32, floor), (name, e_hip, e_t32)

    # (n = 1: BatchNorm over 49 positions of one sample; a pre-activation at rounding distance from 0 may fall on the other side
    # of a ReLU in ANY float32 evaluation -- the floors allow for a few such elements among the batch's thousands)
    check("logp", ph, p32, p64, 5e-6)
    check("v", vh, v32, v64, 5e-6)
    assert abs(lh - l64) <= max(4 * abs(l32 - l64), 5e-6 * abs(l64))
    for k in g64:
        if k.endswith("conv0.bias") or k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            # a conv bias in front of a training-mode BatchNorm has gradient exactly 0; float paths return rounding noise
            scale = float(g64[k.replace("bias", "weight")].abs().max())
            assert float(gh[k].abs().max()) <= 1e-4 * scale + 1e-12, k
            continue
        check(k, gh[k], g32[k], g64[k], 2e-4 if n * (rows + 1) * (cols + 1) > 5000 else 2e-5)
    for k in s64:
        check(k, sh[k], s32[k], s64[k], 2e-6)


def test_hip_heads_are_the_default_and_fall_back_cleanly():
    """training_forward picks the whole-network path for the shipped shape, the tower-only path when the heads differ."""
    from dotsboxesaz_amd import nn as dnn, train as T, train_tower
    torch.manual_seed(0)
    m = dnn.ResNetZero(dnn.resnet_params(3, 3, 64, 1)).cuda().train(True)
    x = (torch.rand(8, 3, 4, 4, device="cuda") < 0.5).float()
    assert train_tower.net_supported(m, x)
    p, v = T.training_forward(m, x)
    assert type(p.grad_fn).__name__.startswith("_NetFn")
    m2 = dnn.ResNetZero(dnn.resnet_params(3, 3, 64, 1, head_channels=8)).cuda().train(True)
    assert not train_tower.net_supported(m2, x) and train_tower.supported(m2, x)
    p2, v2 = T.training_forward(m2, x)
    assert not type(p2.grad_fn).__name__.startswith("_NetFn")
    with pytest.raises(RuntimeError):
        T.training_forward(m2, x, hip_heads=True)
    m.eval()
    assert not type(T.training_forward(m, x)[0].grad_fn).__name__.startswith("_NetFn")


def test_net_backward_needs_its_forward():
    from dotsboxesaz_amd import nn as dnn, train as T, train_tower
    torch.manual_seed(1)
    m = dnn.ResNetZero(dnn.resnet_params(3, 3, 64, 1)).cuda().train(True)
    x = (torch.rand(8, 3, 4, 4, device="cuda") < 0.5).float()
    p1, v1 = T.training_forward(m, x)
    p2, v2 = T.training_forward(m, x)            # the handle now holds the second pass
    with pytest.raises(train_tower.TrainerError):
        (p1.sum() + v1.sum()).backward()
    (p2.sum() + v2.sum()).backward()
    assert m.resnet.conv0.weight.grad is not None and torch.isfinite(m.resnet.conv0.weight.grad).all()

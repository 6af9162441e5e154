"""Training-mode residual tower on HIP (csrc/train.hip, dbaz_trainer_*) against torch autograd.

Ground truth = the same blocks (oracle/nn_ref._Block: nn.py:33-58 restated) evaluated by torch in FLOAT64 on the CPU in
train mode; the HIP path works in f32 (f16x3 MFMA products, f64 statistics), so it is compared with a tolerance relative
to each tensor's largest magnitude and, beside it, with torch's own float32 result: the HIP path must be as close to the
float64 truth as torch float32 is (within a small factor).  Parity status: torch restatement of the reference's blocks --
the reference's own training goldens (tests/golden/train_*.npz) are covered by test_hip_train_step below through
train.train()."""
import copy

import numpy as np
import pytest
import torch

from oracle import nn_ref

pytestmark = pytest.mark.gpu


def make_blocks(nb, seed):
    torch.manual_seed(seed)
    blocks = torch.nn.Sequential(*[nn_ref._Block(64, 3) for _ in range(nb)])
    g = torch.Generator().manual_seed(seed + 1)
    for m in blocks.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(64, generator=g) + 0.5
            m.bias.data = torch.randn(64, generator=g) * 0.2
            m.running_mean.data = torch.randn(64, generator=g) * 0.1
            m.running_var.data = torch.rand(64, generator=g) + 0.5
    return blocks


def run_torch(blocks, x, gout, dtype):
    b = copy.deepcopy(blocks).to(dtype)
    b.train(True)
    xx = x.to(dtype).clone().requires_grad_(True)
    out = b(xx)
    out.backward(gout.to(dtype))
    grads = {k: p.grad.double() for k, p in b.named_parameters()}
    stats = {k: v.double() for k, v in b.state_dict().items() if "running" in k}
    return out.detach().double(), xx.grad.double(), grads, stats


def run_hip(blocks, x, gout):
    from dotsboxesaz_amd import train_tower

    class M:  # the container shape train_tower expects: model.resnet.resblocks / conv0
        pass
    b = copy.deepcopy(blocks).cuda()
    b.train(True)
    m = M()
    m.resnet = M()
    m.resnet.resblocks = b
    m.resnet.conv0 = torch.nn.Conv2d(3, 64, 3)
    xx = x.cuda().clone().requires_grad_(True)
    assert train_tower.supported(m, xx)
    out = train_tower.resblocks_forward(m, xx)
    out.backward(gout.cuda())
    torch.cuda.synchronize()
    grads = {k: p.grad.double().cpu() for k, p in b.named_parameters()}
    stats = {k: v.double().cpu() for k, v in b.state_dict().items() if "running" in k}
    nbt = [int(v) for k, v in b.state_dict().items() if "num_batches_tracked" in k]
    return out.detach().double().cpu(), xx.grad.double().cpu(), grads, stats, nbt


def relu_margins(blocks, x):
    """per sample: the smallest |pre-activation| any ReLU of the float64 reference sees.  An element within rounding
    distance of 0 can come out on the other side of the ReLU in ANY float32 evaluation (torch's included); its gradient
    mask then differs and the backward pass of that sample is not comparable."""
    b = copy.deepcopy(blocks).double()
    b.train(True)
    xx = x.double()
    m = torch.full((x.shape[0],), float("inf"), dtype=torch.float64)
    with torch.no_grad():
        for blk in b:
            pre1 = blk.bn1(blk.conv1(xx))
            pre2 = blk.bn2(blk.conv2(torch.relu(pre1))) + xx
            m = torch.minimum(m, torch.minimum(pre1.abs().flatten(1).min(1)[0], pre2.abs().flatten(1).min(1)[0]))
            xx = torch.relu(pre2)
    return m


UNSAFE = 1.5e-6


def rel(a, ref):
    return float((a - ref).abs().max() / max(float(ref.abs().max()), 1e-30))


@pytest.mark.parametrize("board,nb,n", [((6, 6), 2, 37), ((3, 3), 1, 70), ((9, 9), 1, 5), ((2, 3), 1, 9), ((6, 6), 3, 256)])
def test_tower_forward_backward_vs_float64(board, nb, n):
    H, W = board[0] + 1, board[1] + 1
    blocks = make_blocks(nb, 7 * nb + n)
    big = n * H * W * nb > 20000
    for seed in range(n, n + 40):
        g = torch.Generator().manual_seed(seed)
        x = torch.relu(torch.randn(n, 64, H, W, generator=g))      # the tower input is a post-ReLU activation
        gout = torch.randn(n, 64, H, W, generator=g) * 1e-3        # gradients are small numbers: exercises the dynamic scaling
        safe = relu_margins(blocks, x) >= UNSAFE
        if big or bool(safe.all()):
            break
    # small cases: inputs without a pre-activation at rounding distance from 0, everything compared tightly; the big case
    # (millions of ReLU inputs) always has a few: their samples are left out of grad_x and the parameter gradients, which
    # sum over all samples, are compared loosely
    assert big or bool(safe.all())
    loose = not bool(safe.all())
    o64, gx64, gr64, st64 = run_torch(blocks, x, gout, torch.float64)
    o32, gx32, gr32, st32 = run_torch(blocks, x, gout, torch.float32)
    oh, gxh, grh, sth, nbt = run_hip(blocks, x, gout)
    assert nbt == [1] * (2 * nb)

    def check(name, hip, t32, t64, floor=2e-6):
        e_hip, e_t32 = rel(hip, t64), rel(t32, t64)
        assert e_hip <= max(4 * e_t32, floor), (name, e_hip, e_t32)

    check("out", oh, o32, o64)
    # (a flipped element elsewhere in the batch still reaches the safe samples through the batch statistics: ~1/M)
    check("grad_x", gxh[safe], gx32[safe], gx64[safe], 5e-5 if loose else 2e-6)
    for k in gr64:
        if loose and not (k.endswith("conv1.bias") or k.endswith("conv2.bias")):
            assert rel(grh[k], gr64[k]) < 2e-2, k
            continue
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            # a conv bias in front of a training-mode BatchNorm has gradient exactly 0; float paths return rounding noise
            scale = float(gr64[k.replace("bias", "weight")].abs().max())
            assert float(grh[k].abs().max()) <= 1e-4 * scale + 1e-12, k
            continue
        check(k, grh[k], gr32[k], gr64[k])
    for k in st64:
        check(k, sth[k], st32[k], st64[k])


def test_tower_small_gradients_keep_precision():
    """Gradients of 1e-12 magnitude (far below f16's range) come back with f32-grade relative precision."""
    blocks = make_blocks(1, 3)
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(16, 64, 7, 7, generator=g))
    gout = torch.randn(16, 64, 7, 7, generator=g) * 1e-12
    o64, gx64, gr64, _ = run_torch(blocks, x, gout, torch.float64)
    oh, gxh, grh, _, _ = run_hip(blocks, x, gout)
    assert rel(gxh, gx64) < 1e-5
    for k in gr64:
        if not k.endswith("bias") or "bn" in k:
            assert rel(grh[k], gr64[k]) < 1e-5, k


def test_trainer_errors():
    from dotsboxesaz_amd import train_tower
    with pytest.raises(train_tower.TrainerError):
        train_tower.TowerTrainer(6, 6, 32, 2, 16)          # 64 channels only
    t = train_tower.TowerTrainer(6, 6, 64, 1, 8)
    x = torch.zeros(9, 64, 7, 7, device="cuda")
    w = [torch.zeros(64, 64, 3, 3, device="cuda")] * 2
    v = [torch.zeros(64, device="cuda")] * 2
    with pytest.raises(train_tower.TrainerError):
        t.forward(x, w, v, v, v, v, v)                      # batch beyond max_batch
    with pytest.raises(train_tower.TrainerError):
        t.backward(x[:8], v, w, v, v, v)                    # no forward pass held
    t.close()


def test_one_handle_holds_one_forward_pass():
    """A trainer handle keeps the activations of ONE forward pass.  A second forward through the same handle before the first
    graph's backward must make that backward RAISE (it used to return the gradients of the wrong batch); two models of one shape
    get a handle each; a caller that needs two live graphs passes its own trainers."""
    from dotsboxesaz_amd import train_tower

    class M:
        pass

    def model(seed):
        m = M()
        m.resnet = M()
        m.resnet.resblocks = make_blocks(1, seed).cuda()
        m.resnet.resblocks.train(True)
        return m
    g = torch.Generator().manual_seed(1)
    xa = torch.relu(torch.randn(6, 64, 7, 7, generator=g)).cuda().requires_grad_(True)
    xb = torch.relu(torch.randn(9, 64, 7, 7, generator=g)).cuda().requires_grad_(True)
    xc = torch.relu(torch.randn(4, 64, 7, 7, generator=g)).cuda().requires_grad_(True)
    m1, m2 = model(3), model(4)
    train_tower._trainers.clear()
    ya = train_tower.resblocks_forward(m1, xa)
    yc = train_tower.resblocks_forward(m1, xc)      # same model, fits the same handle: the handle now holds pass c
    with pytest.raises(train_tower.TrainerError):
        ya.sum().backward()
    yc.sum().backward()                             # the pass the handle holds still differentiates
    assert xc.grad is not None and torch.isfinite(xc.grad).all()
    # a LARGER batch makes a new handle; the old one is not closed under the graph that still references it
    ya = train_tower.resblocks_forward(m1, xa)
    yb = train_tower.resblocks_forward(m1, xb)
    ya.sum().backward()
    yb.sum().backward()
    assert torch.isfinite(xa.grad).all() and torch.isfinite(xb.grad).all()
    # two models of the same shape: separate handles, both graphs alive
    train_tower._trainers.clear()
    y1 = train_tower.resblocks_forward(m1, xa)
    y2 = train_tower.resblocks_forward(m2, xa)
    assert len(train_tower._trainers) == 2
    (y1.sum() + y2.sum()).backward()
    # two micro-batches of ONE model summed into one loss: a trainer per live graph
    t1 = train_tower.TowerTrainer(6, 6, 64, 1, 9)
    t2 = train_tower.TowerTrainer(6, 6, 64, 1, 9)
    xa2, xb2 = xa.detach().clone().requires_grad_(True), xb.detach().clone().requires_grad_(True)
    ref = copy.deepcopy(m1.resnet.resblocks)
    loss = train_tower.resblocks_forward(m1, xa2, trainer=t1).sum() + train_tower.resblocks_forward(m1, xb2, trainer=t2).sum()
    loss.backward()
    xr = xa.detach().clone().requires_grad_(True)
    ref(xr).sum().backward()
    assert float((xa2.grad - xr.grad).abs().max()) < 1e-3 * float(xr.grad.abs().max())
    t1.close()
    t2.close()
    train_tower._trainers.clear()


def test_unsupported_batchnorm_settings_stay_on_torch():
    """csrc/train.hip hard-codes eps 1e-5 / momentum 0.1 / affine / running statistics: any other BatchNorm setting must not be
    routed to it (train_tower.supported)."""
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd import train_tower
    m = dnn.ResNetZero(dnn.resnet_params(3, 3, 64, 1)).cuda()
    x = torch.zeros(2, 3, 4, 4, device="cuda")
    assert train_tower.supported(m, x)
    m.resnet.resblocks[0].bn1.eps = 1e-3
    assert not train_tower.supported(m, x)
    m.resnet.resblocks[0].bn1.eps = 1e-5
    m.resnet.resblocks[0].bn2.momentum = None
    assert not train_tower.supported(m, x)
    m.resnet.resblocks[0].bn2.momentum = 0.1
    assert train_tower.supported(m, x) and not train_tower.supported(m, x.double()) and not train_tower.supported(m.double(), x)


def test_train_steps_hip_tower_vs_torch_tower():
    """NeuralNetWrapper.train's step (nn.py:203-221: forward, AlphaZeroLoss, backward, SGD momentum + weight decay) on a
    64-channel ResNetZero for a few batches: residual blocks on csrc/train.hip against the same container with the blocks
    on torch autograd.  Same initial weights and batches; losses, every parameter, the running statistics and
    num_batches_tracked must agree to float32 noise."""
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd import train as T
    params = dnn.resnet_params(6, 6, 64, 3)
    torch.manual_seed(3)
    m_hip = dnn.ResNetZero(params).cuda()
    m_ref = copy.deepcopy(m_hip)
    g = torch.Generator().manual_seed(11)
    crit = T.AlphaZeroLoss()
    opts = [torch.optim.SGD(m.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4) for m in (m_hip, m_ref)]
    for step in range(4):
        boards = torch.randint(0, 2, (192, 3, 7, 7), generator=g).float().cuda()
        pi = torch.softmax(torch.randn(192, 98, generator=g), 1).cuda()
        z = (torch.randint(0, 3, (192, 1), generator=g).float() - 1).cuda()
        losses = []
        for m, opt, hip in ((m_hip, opts[0], True), (m_ref, opts[1], False)):
            m.train(True)
            p, v = T.training_forward(m, boards, hip_tower=hip)
            loss, (lpi, lv) = crit(p, v, pi, z)
            loss.backward()
            opt.step()
            opt.zero_grad()
            losses.append((loss.item(), lpi, lv))
        assert np.allclose(losses[0], losses[1], rtol=2e-5, atol=1e-6), (step, losses)
    sd_h, sd_r = m_hip.state_dict(), m_ref.state_dict()
    worst = {}
    for k in sd_r:
        a, b = sd_h[k].double().cpu(), sd_r[k].double().cpu()
        if "num_batches_tracked" in k:
            assert int(a) == int(b) == 4, k
            continue
        worst[k] = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
    bad = {k: v for k, v in worst.items() if v > 1e-4}
    assert not bad, (bad, max(worst.values()))
    # evaluation mode (validation pass of the reference's loop) goes through torch with the running statistics either way
    m_hip.train(False)
    with torch.no_grad():
        p, v = T.training_forward(m_hip, boards)
    assert torch.isfinite(p).all() and torch.isfinite(v).all()
    with pytest.raises(RuntimeError):
        T.training_forward(m_hip, boards, hip_tower=True)   # required, but the model is in eval mode


def test_full_size_step_batch_4096_20_blocks():
    """The reference's training shape (configuration.py: train_batch_size 4096, ResNetZero 20 x 64 on 6x6): output against
    torch float32 on the GPU (ReLU flips cannot occur in the forward pass), gradients loosely (millions of ReLU inputs:
    some sit at rounding distance from 0 in either evaluation), bit-identical results when the same step runs twice."""
    nb, n = 20, 4096
    blocks = make_blocks(nb, 1)
    g = torch.Generator().manual_seed(2)
    x = torch.relu(torch.randn(n, 64, 7, 7, generator=g))
    gout = torch.randn(n, 64, 7, 7, generator=g) * 1e-3
    oh, gxh, grh, sth, nbt = run_hip(blocks, x, gout)
    oh2, gxh2, grh2, _, _ = run_hip(blocks, x, gout)
    assert torch.equal(oh, oh2) and torch.equal(gxh, gxh2) and all(torch.equal(grh[k], grh2[k]) for k in grh)
    b = copy.deepcopy(blocks).cuda()
    b.train(True)
    xx = x.cuda().clone().requires_grad_(True)
    out = b(xx)
    out.backward(gout.cuda())
    o32, gx32 = out.detach().double().cpu(), xx.grad.double().cpu()
    assert rel(oh, o32) < 2e-5

    def l2(a, r):
        return float((a - r).norm() / r.norm())
    # ~500 of the 5e8 ReLU inputs land on different sides of 0 in two float32 evaluations: a few percent of the samples differ
    # visibly, the rest to rounding (measured: 247 samples, median per-sample deviation 1.3e-5, L2 3e-3)
    per_sample = (gxh - gx32).abs().flatten(1).max(1)[0] / gx32.abs().max()
    assert torch.isfinite(gxh).all() and l2(gxh, gx32) < 2e-2 and float(per_sample.median()) < 1e-4
    assert int((per_sample > 1e-4).sum()) < n // 6
    for k, p in b.named_parameters():
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            continue
        assert l2(grh[k], p.grad.double().cpu()) < 2e-2, k
    for k, v in b.state_dict().items():
        if "running" in k:
            assert rel(sth[k], v.double().cpu()) < 1e-5, k


@pytest.mark.parametrize("ch,hw,n,relu", [(3, (7, 7), 257, False), (64, (7, 7), 130, True), (16, (4, 4), 33, True), (16, (10, 10), 9, True), (5, (3, 4), 1, False)])
def test_batch_norm_train_vs_torch_float64(ch, hw, n, relu):
    """dbaz_bn2d_*: BatchNorm2d in training mode (+ ReLU) on NCHW tensors against torch autograd in float64: output, input
    gradient, dgamma / dbeta, running statistics, num_batches_tracked."""
    from dotsboxesaz_amd import train_tower
    g = torch.Generator().manual_seed(ch * 100 + n)
    bn = torch.nn.BatchNorm2d(ch)
    bn.weight.data = torch.rand(ch, generator=g) + 0.5
    bn.bias.data = torch.randn(ch, generator=g) * 0.3
    bn.running_mean.data = torch.randn(ch, generator=g) * 0.1
    bn.running_var.data = torch.rand(ch, generator=g) + 0.5
    x = torch.randn(n, ch, *hw, generator=g) * 2 + 0.5
    dout = torch.randn(n, ch, *hw, generator=g)
    ref = copy.deepcopy(bn).double()
    ref.train(True)
    xr = x.double().clone().requires_grad_(True)
    yr = ref(xr)
    if relu:
        yr = torch.relu(yr)
    yr.backward(dout.double())
    hip = copy.deepcopy(bn).cuda()
    hip.train(True)
    xh = x.cuda().clone().requires_grad_(True)
    yh = train_tower.batch_norm_train(hip, xh, relu=relu)
    yh.backward(dout.cuda())
    torch.cuda.synchronize()
    if n * hw[0] * hw[1] > 1:
        assert rel(yh.detach().double().cpu(), yr.detach()) < 2e-6
        assert rel(xh.grad.double().cpu(), xr.grad) < 2e-5
        assert rel(hip.weight.grad.double().cpu(), ref.weight.grad) < 2e-5 and rel(hip.bias.grad.double().cpu(), ref.bias.grad) < 2e-5
        assert rel(hip.running_mean.double().cpu(), ref.running_mean) < 2e-6 and rel(hip.running_var.double().cpu(), ref.running_var) < 2e-6
    assert int(hip.num_batches_tracked) == 1


def test_exact_f32_weight_gradient_path(monkeypatch):
    """Debug build only (-DDBAZ_DEBUG, DBAZ_LIB=.../libdbaz_hip_debug.so): DBAZ_TRAIN_WGRAD_F32=1 selects k_wgrad
    (v_mfma_f32_16x16x4_f32, exact products) instead of k_wgrad_h3: same gradients.  The release library reads no environment."""
    from dotsboxesaz_amd import train_tower, _lib
    if "debug" not in _lib.load().dbaz_build_info().decode():
        pytest.skip("release build: the exact-f32 weight gradient kernel is an A/B path of the debug build")
    blocks = make_blocks(1, 5)
    g = torch.Generator().manual_seed(9)
    x = torch.relu(torch.randn(21, 64, 7, 7, generator=g))
    gout = torch.randn(21, 64, 7, 7, generator=g) * 1e-2
    _, _, gr64, _ = run_torch(blocks, x, gout, torch.float64)
    train_tower._trainers.clear()
    _, _, gr_h3, _, _ = run_hip(blocks, x, gout)
    monkeypatch.setenv("DBAZ_TRAIN_WGRAD_F32", "1")
    train_tower._trainers.clear()
    _, _, gr_f32, _, _ = run_hip(blocks, x, gout)
    train_tower._trainers.clear()
    for k in ("0.conv1.weight", "0.conv2.weight"):
        assert rel(gr_f32[k], gr64[k]) < 2e-5 and rel(gr_h3[k], gr64[k]) < 2e-5 and rel(gr_f32[k], gr_h3[k]) < 2e-5


@pytest.mark.parametrize("tag", ["t33", "t66"])
def test_golden_reference_resblocks_in_training_mode(tag):
    """tests/golden/train_tower.npz: the REFERENCE's own ResBlock modules (nn.py:33-58) under .train(True), forward + backward in
    float32 on the CPU (gen_golden.gen_train_tower; seeds chosen so that no ReLU input lies within 2e-6 of zero).  The weights
    are regenerated here from the stored seed (checksum compared), the HIP tower must reproduce output, input gradient, every
    parameter gradient (conv weights: every 37th element), running statistics and num_batches_tracked."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_tower.npz"))
    r, c, nb, n, seed = (int(v) for v in G[tag + "_cfg"])
    torch.manual_seed(seed)
    blocks = torch.nn.Sequential(*[nn_ref._Block(64, 3) for _ in range(nb)])
    nn_ref.randomize_bn(blocks, seed + 1)
    assert nn_ref.state_dict_checksum(blocks) == float(G[tag + "_checksum"])
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.relu(torch.randn(n, 64, r + 1, c + 1, generator=g))
    gout = torch.randn(n, 64, r + 1, c + 1, generator=g) * 1e-2
    oh, gxh, grh, sth, nbt = run_hip(blocks, x, gout)
    assert rel(oh, torch.tensor(G[tag + "_out"]).double()) < 2e-6
    assert rel(gxh, torch.tensor(G[tag + "_grad_x"]).double()) < 2e-5
    for k in grh:
        ref = torch.tensor(G[tag + "_g_" + k]).double()
        got = grh[k] if ref.numel() == grh[k].numel() else grh[k].reshape(-1)[::37]
        if k.endswith("conv1.bias") or k.endswith("conv2.bias"):
            assert float(got.abs().max()) <= 1e-4 * float(grh[k.replace("bias", "weight")].abs().max()) + 1e-12, k  # exactly 0 analytically
            continue
        assert rel(got.reshape(ref.shape), ref) < 2e-5, k
    for k, v in sth.items():
        assert rel(v, torch.tensor(G[tag + "_s_" + k]).double()) < 2e-6, k
    assert nbt == [1] * (2 * nb)


def test_hip_alphazero_loss_vs_torch():
    """AlphaZeroLoss (nn.py:131-138) forward + backward as HIP kernels (dbaz_az_loss) against torch autograd of the reference's
    two statements; ragged sizes, an upstream gradient other than 1."""
    from dotsboxesaz_amd import train as T
    g = torch.Generator().manual_seed(5)
    for n, A in ((4096, 98), (37, 32), (1, 200), (515, 98)):
        logits = torch.randn(n, A, generator=g)
        p0 = torch.log_softmax(logits, 1).cuda()
        v0 = torch.tanh(torch.randn(n, 1, generator=g)).cuda()
        pi = torch.softmax(torch.randn(n, A, generator=g) * 2, 1).cuda()
        z = (torch.randint(0, 3, (n, 1), generator=g).float() - 1).cuda()
        res = []
        for hip in (True, False):
            p, v = p0.clone().requires_grad_(True), v0.clone().requires_grad_(True)
            loss, (lpi, lv) = T.AlphaZeroLoss.tensors(p, v, pi, z, hip=hip)
            (loss * 1.7).backward()
            res.append((loss.item(), lpi.item(), lv.item(), p.grad.clone(), v.grad.clone()))
        a, b = res
        for i in range(3):
            assert abs(a[i] - b[i]) <= 2e-6 * max(1.0, abs(b[i])), (n, A, i, a[i], b[i])
        assert torch.allclose(a[3], b[3], rtol=1e-6, atol=1e-9) and torch.allclose(a[4], b[4], rtol=1e-6, atol=1e-9)
        assert a[4].shape == v0.shape
    # the module's forward (python floats, nn.py:136-138) goes through the same kernels on CUDA tensors
    loss, (lpi, lv) = T.AlphaZeroLoss()(p0, v0, pi, z)
    assert isinstance(lpi, float) and abs(float(loss) - (lpi + lv)) < 1e-6


def test_hip_sgd_is_torch_sgd():
    """HipSGD.step() (one HIP launch over all parameter tensors, dbaz_sgd_step) against torch.optim.SGD with momentum and weight decay
    (what nn.py:179 builds from configuration.py:62-66): parameters and momentum buffers after 5 steps, a state_dict written by one
    loaded into the other (the reference's checkpoint format), tensors whose gradient is None, a parameter without momentum."""
    from dotsboxesaz_amd import train as T
    g = torch.Generator().manual_seed(9)
    shapes = [(1,), (7,), (64, 64, 3, 3), (2048,), (2049,), (98, 784), (8,), (3, 5, 7)]
    for mom, wd in ((0.9, 1e-4), (0.0, 0.0), (0.5, 0.0)):
        pa = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
        pb = [p.detach().clone().requires_grad_(True) for p in pa]
        oa = T.HipSGD(pa, lr=3e-2, momentum=mom, weight_decay=wd)
        ob = torch.optim.SGD(pb, lr=3e-2, momentum=mom, weight_decay=wd)
        for step in range(5):
            for i, (a, b) in enumerate(zip(pa, pb)):
                if i == 4 and step % 2 == 1:
                    a.grad = b.grad = None                      # a parameter that got no gradient this step
                    continue
                gr = torch.randn(a.shape, generator=g).cuda()
                a.grad, b.grad = gr.clone(), gr.clone()
            oa.step()
            ob.step()
            if step == 2:                                       # checkpoint exchange in both directions (nn.py:292-313)
                sa, sb = oa.state_dict(), ob.state_dict()
                assert sa["param_groups"][0]["momentum"] == sb["param_groups"][0]["momentum"] and set(sa["state"]) == set(sb["state"])
                oa.load_state_dict(sb)
                ob.load_state_dict(sa)
        for a, b in zip(pa, pb):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7), (mom, wd, a.shape, float((a - b).abs().max()))
        if mom:
            for a, b in zip(pa, pb):
                # (torch's multi-tensor kernels round g + wd p and momentum buf + d in a slightly different order: an ulp of the
                # larger operand, 2e-7 at these magnitudes, survives cancellation)
                assert torch.allclose(oa.state[a]["momentum_buffer"], ob.state[b]["momentum_buffer"], rtol=1e-6, atol=2e-6)
    # CPU parameters: torch's own step
    pc = [torch.randn(5, generator=g).requires_grad_(True)]
    oc = T.HipSGD(pc, lr=0.1, momentum=0.9)
    pc[0].grad = torch.ones(5)
    before = pc[0].detach().clone()
    oc.step()
    assert torch.allclose(pc[0], before - 0.1)


def test_a_handle_goes_with_its_model():
    """coach.train_nn builds a new model every generation: its trainer handle (activations of a whole batch in HBM) must not outlive
    it.  The cache holds models weakly; the device memory of a dead model's handle is returned."""
    import gc
    from dotsboxesaz_amd import nn as dnn, train as T, train_tower
    train_tower._trainers.clear()
    gc.collect()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    x = (torch.rand(256, 3, 7, 7, device="cuda") < 0.4).float()
    for _ in range(3):      # three "generations": a new model each, as coach.py:71 does
        m = dnn.ResNetZero(dnn.resnet_params(6, 6, 64, 4)).cuda().train(True)
        p, v = T.training_forward(m, x)
        (p.sum() + v.sum()).backward()
        torch.cuda.synchronize()
        assert len(train_tower._trainers) == 1
        del m, p, v
        gc.collect()
    assert len(train_tower._trainers) == 0
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    # one handle of this size is ~0.3 GB: nothing of that order may be left behind
    assert free0 - free1 < 64 * 2 ** 20, (free0, free1)

"""Torch fp32 CPU restatement of the reference's policy/value networks.

TEST INFRASTRUCTURE (oracle) -- floating-point reference for the HIP conv
kernels.  Architecture and state_dict key names follow the reference:
  ResNetZero  nn.py:108-122 (ResNet :16-30, ResBlock :33-58, PolicyHead :74-87,
              ValueHead :90-105), shipped config configuration.py:134-156
  SimpleNN    dots_boxes/dots_boxes_nn.py:61-98
  predict     NeuralNetWrapper.predict_sync nn.py:155-160 (eval mode, exp(log p))

Parity status: pinned by tests/golden/nn_*.npz (outputs of the reference's own
modules on committed / seed-regenerated weights).
"""
import torch
from torch import nn
import torch.nn.functional as F


def _conv(cin, cout, k, pad=True, groups=1):
    # nn.py:61-71 (_create_conv_layer, odd kernels): groups = n_groups
    return nn.Conv2d(cin, cout, k, padding=(k - 1) // 2 if pad else 0, groups=groups)


class _Block(nn.Module):
    def __init__(self, ch, k, groups=1):
        super().__init__()
        self.conv1 = _conv(ch, ch, k, groups=groups)
        self.bn1 = nn.BatchNorm2d(ch)
        self.conv2 = _conv(ch, ch, k, groups=groups)
        self.bn2 = nn.BatchNorm2d(ch)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y = y + x
        return F.relu(y)


class _Tower(nn.Module):
    def __init__(self, cin, ch, k, nb, groups=1):
        super().__init__()
        self.conv0 = _conv(cin, ch, 3)  # (nn.py:17: the first conv is never grouped)
        self.bn0 = nn.BatchNorm2d(ch)
        self.resblocks = nn.Sequential(*[_Block(ch, k, groups) for _ in range(nb)])

    def forward(self, x):
        return self.resblocks(F.relu(self.bn0(self.conv0(x))))


class _ValueHead(nn.Module):
    def __init__(self, cin, inner, fc_in, fc_inner):
        super().__init__()
        self.conv0 = nn.Conv2d(cin, inner, kernel_size=(1, 1))
        self.bn0 = nn.BatchNorm2d(inner)
        self.fc0 = nn.Linear(fc_in, fc_inner)
        self.fc1 = nn.Linear(fc_inner, 1)

    def forward(self, x):
        x = F.relu(self.bn0(self.conv0(x)))
        x = F.relu(self.fc0(x.reshape(x.size(0), -1)))
        return torch.tanh(self.fc1(x))


class _PolicyHead(nn.Module):
    def __init__(self, cin, inner, fc_in, nb_actions):
        super().__init__()
        self.conv0 = nn.Conv2d(cin, inner, kernel_size=(1, 1))
        self.bn0 = nn.BatchNorm2d(inner)
        self.fc = nn.Linear(fc_in, nb_actions)

    def forward(self, x):
        x = F.relu(self.bn0(self.conv0(x)))
        return F.log_softmax(self.fc(x.reshape(x.size(0), -1)), dim=1)


class ResNetZeroRef(nn.Module):
    """Same module tree / registration order as the reference's ResNetZero, so a
    reference state_dict loads with strict=True and torch.manual_seed(s) yields the
    same random init."""

    def __init__(self, rows, cols, channels=64, blocks=20, kernel=3, head_channels=16, value_fc=8,
                 in_channels=3, n_groups=1):
        super().__init__()
        H, W = rows + 1, cols + 1
        self.cfg = dict(rows=rows, cols=cols, channels=channels, blocks=blocks, kernel=kernel,
                        head_channels=head_channels, value_fc=value_fc, in_channels=in_channels, n_groups=n_groups)
        self.bn_input = nn.BatchNorm2d(in_channels)
        self.resnet = _Tower(in_channels, channels, kernel, blocks, n_groups)
        self.value_head = _ValueHead(channels, head_channels, head_channels * H * W, value_fc)
        self.policy_head = _PolicyHead(channels, head_channels, head_channels * H * W, 2 * H * W)

    def forward(self, x):
        x = self.resnet(self.bn_input(x))
        return self.policy_head(x), self.value_head(x)


class SimpleNNRef(nn.Module):
    """dots_boxes_nn.py:61-98 -- 3x3 boards only; BN follows ReLU."""

    def __init__(self, n_ch=256):
        super().__init__()
        self.conv0 = nn.Conv2d(3, n_ch, 3, padding=1)
        self.bn0 = nn.BatchNorm2d(n_ch)
        self.conv1 = nn.Conv2d(n_ch, n_ch, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(n_ch)
        self.conv2 = nn.Conv2d(n_ch, n_ch, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(n_ch)
        self.conv3 = nn.Conv2d(n_ch, n_ch, 3, padding=1)
        self.bn3 = nn.BatchNorm2d(n_ch)
        self.conv4 = nn.Conv2d(n_ch, n_ch, 3, padding=0)
        self.bn4 = nn.BatchNorm2d(n_ch)
        self.fc0 = nn.Linear(1024, 512)
        self.bn_fc0 = nn.BatchNorm1d(512)
        self.fc1 = nn.Linear(512, 256)
        self.bn_fc1 = nn.BatchNorm1d(256)
        self.value_fc = nn.Linear(256, 1)
        self.policy_fc = nn.Linear(256, 32)

    def forward(self, x):
        for i in range(5):
            x = getattr(self, "bn%d" % i)(F.relu(getattr(self, "conv%d" % i)(x)))
        x = x.reshape(x.size(0), -1)
        x = self.bn_fc0(F.relu(self.fc0(x)))
        x = self.bn_fc1(F.relu(self.fc1(x)))
        return F.log_softmax(self.policy_fc(x), dim=1), torch.tanh(self.value_fc(x))


def randomize_bn(model, seed):
    """Give every BatchNorm non-trivial affine + running statistics (a fresh module
    has mean 0 / var 1 / gamma 1 / beta 0, which would hide BN bugs).  Deterministic
    in module registration order; applied identically to reference and restated nets."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                n = m.num_features
                m.running_mean.copy_(torch.randn(n, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(n, generator=g) * 0.5 + 0.75)
                m.weight.copy_(torch.rand(n, generator=g) * 0.5 + 0.75)
                m.bias.copy_(torch.randn(n, generator=g) * 0.1)
    return model


def trained_like_(model, X, seed, gamma_decades=3.0, var_decades=3.0, mean_sigmas=2.0):
    """Give a ResNetZeroRef the STATISTICS of a trained network, layer by layer on the probe batch X (float32 [n,3,H,W]), in place:
      * every conv's output channels are rescaled by log-uniform factors so that the variance a BatchNorm sees spreads over
        `var_decades` decades inside ONE layer, and the layer's running_var is set to the variance actually observed on the probe
        (what training does), running_mean to the observed mean shifted by up to +-mean_sigmas standard deviations;
      * gamma is log-uniform over `gamma_decades` decades (geometric mean 0.5), beta ~ 0.3 gamma N(0,1);
      * the residual stream is left to grow over the blocks as it does (it is NOT renormalised).
    The folded per-channel weight scale gamma / sigma then spreads over >= gamma_decades decades per layer and the activations over
    several decades between channels: what the f16x3 operand format (one power-of-two scale per layer) has to survive.
    Returns the per-layer maxima of |activation| on the probe (conv0 output first)."""
    g = torch.Generator().manual_seed(seed)
    maxima = []

    def logu(n, decades, centre=1.0):
        return centre * torch.pow(10.0, (torch.rand(n, generator=g) - 0.5) * decades)

    def fit(conv, bn, x, relu_after=True, skip=None):
        n = conv.out_channels
        s = logu(n, var_decades / 2.0)  # variance spreads over var_decades decades
        conv.weight.mul_(s.view(-1, 1, 1, 1))
        conv.bias.mul_(s)
        y = conv(x)
        mu, var = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=False)
        bn.running_var.copy_(var.clamp_min(1e-12))
        bn.running_mean.copy_(mu + (torch.rand(n, generator=g) * 2 - 1) * mean_sigmas * var.sqrt())
        bn.weight.copy_(logu(n, gamma_decades, 0.5))
        bn.bias.copy_(0.3 * bn.weight * torch.randn(n, generator=g))
        z = bn(y)
        if skip is not None:
            z = z + skip
        return F.relu(z) if relu_after else z

    model.train(False)
    with torch.no_grad():
        x = torch.as_tensor(X, dtype=torch.float32)
        mu, var = x.mean(dim=(0, 2, 3)), x.var(dim=(0, 2, 3), unbiased=False)
        model.bn_input.running_mean.copy_(mu)
        model.bn_input.running_var.copy_(var.clamp_min(1e-6))
        model.bn_input.weight.copy_(logu(x.shape[1], 1.0))
        model.bn_input.bias.copy_(0.2 * torch.randn(x.shape[1], generator=g))
        x = model.bn_input(x)
        r = model.resnet
        x = fit(r.conv0, r.bn0, x)
        maxima.append(float(x.abs().max()))
        for blk in r.resblocks:
            y = fit(blk.conv1, blk.bn1, x)
            maxima.append(float(y.abs().max()))
            x = fit(blk.conv2, blk.bn2, y, skip=x)
            maxima.append(float(x.abs().max()))
        for head in (model.policy_head, model.value_head):
            fit(head.conv0, head.bn0, x)
            # heads: moderate gamma so that the logits stay O(1..10) (a trained policy is peaked, not saturated)
            head.bn0.weight.copy_(logu(head.bn0.num_features, 1.0, 0.7))
    return maxima


def predict_sync(model, X):
    """NeuralNetWrapper.predict_sync (nn.py:155-160): eval mode, float32 in,
    (softmax p [n,A], tanh v [n,1]) numpy out."""
    model.train(False)
    with torch.no_grad():
        x = torch.tensor(X, dtype=torch.float32)
        p, v = model.forward(x)
        return torch.exp(p).cpu().numpy(), v.cpu().numpy()


def state_dict_checksum(model):
    s = 0.0
    for _, t in model.state_dict().items():
        s += float(t.double().abs().sum())
    return s

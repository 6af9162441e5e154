"""Torch fp32 restatement (oracle/nn_ref.py) vs golden outputs of the reference's
ResNetZero / SimpleNN / NeuralNetWrapper.predict_sync -- rows N1-N3."""
import numpy as np
import pytest
import torch

from oracle import nn_ref

TOL = 1e-4  # north-star tolerance for policy/value (fp32)


def _small(g, tag):
    cfg = [int(x) for x in g[tag + "_cfg"]]
    r, c, ch, nb, hc, vf = cfg[:6]
    m = nn_ref.ResNetZeroRef(r, c, ch, nb, 3, hc, vf, n_groups=cfg[6] if len(cfg) > 6 else 1)
    sd = {k[len(tag) + 3:]: torch.tensor(g[k]) for k in g.files if k.startswith(tag + "_w_")}
    m.load_state_dict(sd, strict=True)  # reference key names load unchanged
    return m


# groups33 / groups23: the reference's n_groups option (grouped 3x3 convs in the blocks, nn.py:33-47,61-71)
@pytest.mark.parametrize("tag", ["small33", "small66", "small23", "groups33", "groups23"])
def test_committed_weights(golden_nn, tag):
    g = golden_nn
    torch.set_num_threads(1)
    m = _small(g, tag)
    p, v = nn_ref.predict_sync(m, g[tag + "_X"])
    assert p.shape == g[tag + "_p"].shape and v.shape == g[tag + "_v"].shape and v.shape[1] == 1
    assert np.abs(p - g[tag + "_p"]).max() < 1e-6
    assert np.abs(v - g[tag + "_v"]).max() < 1e-6
    assert np.allclose(p.sum(1), 1.0, atol=1e-5)


def test_state_dict_keys_match_reference(golden_nn):
    g = golden_nn
    keys = sorted(k[len("small33_w_"):] for k in g.files if k.startswith("small33_w_"))
    m = _small(g, "small33")
    assert keys == sorted(m.state_dict().keys())
    full = nn_ref.ResNetZeroRef(3, 3)
    assert len(full.state_dict()) == 312


@pytest.mark.parametrize("tag,rows,cols", [("full33", 3, 3), ("full66", 6, 6), ("full99", 9, 9)])
def test_full_size_seeded(golden_nn, tag, rows, cols):
    g = golden_nn
    torch.set_num_threads(1)
    torch.manual_seed(0)
    m = nn_ref.ResNetZeroRef(rows, cols)
    nn_ref.randomize_bn(m, 3)
    cs = nn_ref.state_dict_checksum(m)
    if abs(cs - float(g[tag + "_checksum"])) > 1e-6 * cs:
        pytest.skip("torch RNG stream differs on this host; seeded weights not reproducible")
    p, v = nn_ref.predict_sync(m, g[tag + "_X"])
    assert np.abs(p - g[tag + "_p"]).max() < TOL
    assert np.abs(v - g[tag + "_v"]).max() < TOL


def test_simple_nn_seeded(golden_nn):
    g = golden_nn
    torch.set_num_threads(1)
    torch.manual_seed(0)
    m = nn_ref.SimpleNNRef()
    nn_ref.randomize_bn(m, 3)
    cs = nn_ref.state_dict_checksum(m)
    if abs(cs - float(g["simple_checksum"])) > 1e-6 * cs:
        pytest.skip("torch RNG stream differs on this host")
    p, v = nn_ref.predict_sync(m, g["simple_X"])
    assert np.abs(p - g["simple_p"]).max() < TOL
    assert np.abs(v - g["simple_v"]).max() < TOL


def test_restated_blocks_in_training_mode_match_reference_goldens():
    """tests/golden/train_tower.npz (the reference's ResBlock modules under .train(True), gen_golden.gen_train_tower): the
    restated blocks of oracle/nn_ref.py regenerate the same weights from the seed and reproduce forward and backward bit for
    bit on the CPU -- they are what the HIP training tower is compared with where no golden exists."""
    import os
    import numpy as np
    import torch
    from oracle import nn_ref
    torch.set_num_threads(1)
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_tower.npz"))
    for tag in ("t33", "t66"):
        r, c, nb, n, seed = (int(v) for v in G[tag + "_cfg"])
        torch.manual_seed(seed)
        blocks = torch.nn.Sequential(*[nn_ref._Block(64, 3) for _ in range(nb)])
        nn_ref.randomize_bn(blocks, seed + 1)
        assert nn_ref.state_dict_checksum(blocks) == float(G[tag + "_checksum"])
        g = torch.Generator().manual_seed(seed + 2)
        x = torch.relu(torch.randn(n, 64, r + 1, c + 1, generator=g)).requires_grad_(True)
        gout = torch.randn(n, 64, r + 1, c + 1, generator=g) * 1e-2
        blocks.train(True)
        y = blocks(x)
        y.backward(gout)
        assert np.array_equal(y.detach().numpy(), G[tag + "_out"]) and np.array_equal(x.grad.numpy(), G[tag + "_grad_x"])
        for k, p in blocks.named_parameters():
            gnp = p.grad.numpy()
            assert np.array_equal(gnp if gnp.size <= 64 else gnp.ravel()[::37], G[tag + "_g_" + k]), k
        for k, v in blocks.state_dict().items():
            if "running" in k or "num_batches" in k:
                assert np.array_equal(v.numpy(), G[tag + "_s_" + k]), k


def test_mirror_container_accepts_n_groups(golden_nn):
    """dotsboxesaz_amd.nn.ResNetZero(params) with n_groups > 1 keeps the reference's state_dict keys and shapes (the grouped golden
    loads with strict=True), and the engine-side expansion of a grouped weight is the dense block-diagonal matrix."""
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd.engine import _dense_from_grouped
    g = golden_nn
    r, c, ch, nb, hc, vf, ng = [int(x) for x in g["groups33_cfg"]]
    m = dnn.ResNetZero(dnn.resnet_params(r, c, ch, nb, hc, vf, n_groups=ng))
    sd = {k[len("groups33_w_"):]: torch.tensor(g[k]) for k in g.files if k.startswith("groups33_w_")}
    m.load_state_dict(sd, strict=True)
    w = g["groups33_w_resnet.resblocks.0.conv1.weight"]
    d = _dense_from_grouped(w, ch)
    x = torch.randn(3, ch, 4, 4)
    ref = torch.nn.functional.conv2d(x, torch.tensor(w), padding=1, groups=ng)
    assert torch.equal(torch.nn.functional.conv2d(x, torch.tensor(d), padding=1), ref) or \
        float((torch.nn.functional.conv2d(x, torch.tensor(d), padding=1) - ref).abs().max()) < 1e-6
    with pytest.raises(NotImplementedError):
        dnn.ResNetZero(dnn.resnet_params(r, c, 16, nb, hc, vf, n_groups=3))

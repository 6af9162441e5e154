"""Torch fp32 restatement (oracle/nn_ref.py) vs golden outputs of the reference's
ResNetZero / SimpleNN / NeuralNetWrapper.predict_sync -- rows N1-N3."""
import numpy as np
import pytest
import torch

from oracle import nn_ref

TOL = 1e-4  # north-star tolerance for policy/value (fp32)


def _small(g, tag):
    r, c, ch, nb, hc, vf = [int(x) for x in g[tag + "_cfg"]]
    m = nn_ref.ResNetZeroRef(r, c, ch, nb, 3, hc, vf)
    sd = {k[len(tag) + 3:]: torch.tensor(g[k]) for k in g.files if k.startswith(tag + "_w_")}
    m.load_state_dict(sd, strict=True)  # reference key names load unchanged
    return m


@pytest.mark.parametrize("tag", ["small33", "small66", "small23"])
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

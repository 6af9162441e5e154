"""Optimizer-step plumbing (dotsboxesaz_amd/train.py) against the reference's recorded training run
(tests/golden/train.npz, section (d) of gen_golden.gen_train): AlphaZeroLoss on a fixed batch, then
NeuralNetWrapper.train for generation 1 (2 epochs, SGD momentum 0.9, wd 1e-4) starting from the
generation-0 checkpoint.  CPU torch, one thread, same seeds: weights, scalars and the checkpoint
layout must reproduce.  The symmetry callable used here is the ORACLE's (test infrastructure): the
product's symmetries are HIP-only and covered by tests/test_hip_train_data.py."""
import os
import random

import numpy as np
import pytest
import torch

from dotsboxesaz_amd import nn as dnn
from dotsboxesaz_amd import train as T
from oracle import train_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train.npz"))


class HostDataset(torch.utils.data.Dataset):
    """utils.HDFStoreDataset's item contract (utils/utils.py:85-91) over host arrays."""

    def __init__(self, f, p, v):
        self.features, self.policy, self.value = f, p, v

    def __len__(self):
        return self.features.shape[0]

    def __getitem__(self, i):
        return self.features[i], self.policy[i], np.asarray([self.value[i]])


class OracleSymmetries:
    def __call__(self, boards, pi):
        sym = random.randint(0, 7)
        b, p = train_ref.apply_symmetry(boards.numpy(), pi.numpy(), sym)
        return torch.tensor(np.ascontiguousarray(b)), torch.tensor(np.ascontiguousarray(p))


class Writer:
    def __init__(self):
        self.scalars = []

    def add_scalar(self, tag, v, i):
        self.scalars.append((tag, float(v), int(i)))

    def add_scalars(self, tag, d, i):
        for k, v in sorted(d.items()):
            self.scalars.append((tag + "/" + k, float(v), int(i)))


def _model_with(sd_prefix):
    params = dnn.resnet_params(3, 3, 16, 2, 4, 8)
    model = dnn.ResNetZero(params)
    sd = {k[len(sd_prefix):]: torch.tensor(G[k]) for k in G.files if k.startswith(sd_prefix)}
    model.load_state_dict(sd)
    return model, params


def test_alphazero_loss_matches_reference():
    torch.set_num_threads(1)
    model, _ = _model_with("tr_sd0/")
    model.train(False)
    p, v = T.training_forward(model, torch.tensor(G["ld_boards"][0]))
    loss, (lpi, lv) = T.AlphaZeroLoss()(p, v, torch.tensor(G["ld_pi"][0]), torch.tensor(G["ld_z"][0]))
    assert np.array_equal(np.array([loss.item(), lpi, lv]), G["tr_loss_eval"])


def test_train_generation_1_reproduces_reference(tmp_path):
    torch.set_num_threads(1)
    model0, params = _model_with("tr_sd0/")
    params["nn"]["chkpts_filename"] = str(tmp_path / "model_gen{}.pt")
    params["nn"]["train_params"] = {"nb_epochs": 2, "train_batch_size": 16, "val_batch_size": 16, "lr": 1e-2,
                                    "optimizer_params": {"momentum": 0.9, "weight_decay": 1e-4},
                                    "symmetries": OracleSymmetries()}
    opt0 = torch.optim.SGD(model0.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    T.save_checkpoint(params["nn"]["chkpts_filename"].format(0), model0, opt0, 7)
    ds = HostDataset(G["ds_train_avg_features"], G["ds_train_avg_policy"], G["ds_train_avg_value"])
    val = HostDataset(ds.features[:40], ds.policy[:40], ds.value[:40])
    model = dnn.ResNetZero(params)          # fresh container; train() resumes from the generation-0 checkpoint
    w = Writer()
    seed_t, seed_r = int(G["tr_cfg"][0]), int(G["tr_cfg"][1])
    torch.manual_seed(seed_t)
    random.seed(seed_r)
    last = T.train(model, params, ds, val, w, 1, device="cpu")
    assert last == int(G["tr_last_batch_idx"][0]) == 7 + 2 * (len(ds) // 16)
    assert [t for t, _, _ in w.scalars] == list(G["tr_scalar_tags"])
    assert np.array_equal(np.array([i for _, _, i in w.scalars]), G["tr_scalar_steps"])
    assert np.allclose(np.array([v for _, v, _ in w.scalars]), G["tr_scalar_vals"], rtol=0, atol=1e-6)
    ck = torch.load(params["nn"]["chkpts_filename"].format(1), map_location="cpu", weights_only=True)
    assert sorted(ck.keys()) == list(G["tr_ck_keys"])
    worst = 0.0
    for k, v in ck["model_dict"].items():
        ref = G["tr_sd1/" + k]
        assert v.shape == ref.shape
        worst = max(worst, float(np.max(np.abs(v.numpy().astype(np.float64) - ref))) if ref.size else 0.0)
    assert worst <= 1e-6, worst
    with pytest.raises(ValueError):         # generation 3 needs the generation-2 checkpoint (nn.py:304-305)
        T.train(dnn.ResNetZero(params), params, ds, None, Writer(), 3, device="cpu")


def test_lr_scheduler_and_window():
    s = T.GenerationLrScheduler({0: 1e-2, 20: 1e-3, 50: 1e-4})
    assert [s(g) for g in (0, 19, 20, 49, 50, 80)] == [1e-2, 1e-2, 1e-3, 1e-3, 1e-4, 1e-4]
    # coach.py:148-149
    assert [T.window_where(g) for g in (0, 3, 4, 6, 10, 36, 40)] == [0, 0, 0, 1, 3, 16, 20]


def test_replay_store_drops_what_lies_below_the_window():
    """The training window's lower edge (coach.py:148-149) only moves forward; ReplayStore.drop_before releases the generations below
    it (host logic only: no GPU)."""
    import torch
    from dotsboxesaz_amd import train as T
    from dotsboxesaz_amd.train_data import ReplayStore
    edges = [T.window_where(g) for g in range(0, 80)]
    assert all(b >= a for a, b in zip(edges, edges[1:])) and edges[0] == 0 and max(g - e for g, e in enumerate(edges)) == 20
    st = ReplayStore(None)
    np.random.seed(0)
    for g in range(30):
        st.add_generation(g, torch.zeros((10 + g, 8), dtype=torch.uint8))
        dropped = st.drop_before(T.window_where(g))
        assert dropped in (0, 1)
        assert [c["generation"] for c in st.chunks] == list(range(T.window_where(g), g + 1))
    assert len(st.chunks) <= 21

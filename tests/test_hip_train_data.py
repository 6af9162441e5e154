"""Training data path on the GPU through the C ABI: dbaz_dataset_* / dbaz_symmetry_apply
vs the reference's golden vectors (tests/golden/train.npz) and the numpy oracle (oracle/train_ref.py).
Everything here is integer/permutation work or float64 means cast to float32: the bar is bit-exact."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train.npz"))

ROWMETA = np.dtype([("game_idx", "<i4"), ("move_idx", "<i2"), ("move", "<i2"), ("played", "<i2"), ("max_deepness", "<i2"),
                    ("tree_size", "<i4"), ("terminal_count", "<i4"), ("q_value", "<f4"), ("player", "i1"), ("z", "i1"),
                    ("pad", "<i2")])


def pack_rows(x, visits, z, game_idx=None, move_idx=None):
    """numpy restatement of the packed replay row (RowMeta 28 B | x i16[3HW] | visits i32[A], 8-B padded)."""
    n, F = x.shape
    A = visits.shape[1]
    rb = (28 + 2 * F + 4 * A + 7) // 8 * 8
    rows = np.zeros((n, rb), dtype=np.uint8)
    meta = np.zeros(n, dtype=ROWMETA)
    meta["z"] = z
    meta["game_idx"] = 0 if game_idx is None else game_idx
    meta["move_idx"] = 0 if move_idx is None else move_idx
    rows[:, :28] = meta.view(np.uint8).reshape(n, 28)
    rows[:, 28:28 + 2 * F] = np.ascontiguousarray(x.astype("<i2")).view(np.uint8).reshape(n, 2 * F)
    rows[:, 28 + 2 * F:28 + 2 * F + 4 * A] = np.ascontiguousarray(visits.astype("<i4")).view(np.uint8).reshape(n, 4 * A)
    return rows


def _engine(rows, cols):
    from dotsboxesaz_amd.engine import Engine
    return Engine(rows, cols, 4, mcts_num_read=10, evaluator="formula", nodes_per_slot=64)


@pytest.mark.parametrize("board", [(3, 3), (6, 6)])
def test_symmetry_apply_matches_reference(board):
    import torch
    e = _engine(*board)
    k = "sym%d%d_" % board
    b = torch.tensor(G[k + "boards"]).cuda()
    p = torch.tensor(G[k + "pol"]).cuda()
    for sym in range(8):
        bo, po = e.symmetry_apply(sym, b, p)
        assert np.array_equal(bo.cpu().numpy(), G[k + "boards_out%d" % sym])
        assert np.array_equal(po.cpu().numpy(), G[k + "pol_out%d" % sym])
        _, po2 = e.symmetry_apply(sym, None, p)      # policies only
        assert np.array_equal(po2.cpu().numpy(), G[k + "pol_out%d" % sym])
    e.close()


def test_symmetry_on_non_square_board():
    import torch
    from dotsboxesaz_amd._lib import DbazError
    from oracle import train_ref as T
    e = _engine(2, 3)
    rs = np.random.RandomState(0)
    b = rs.rand(6, 3, 3, 4).astype(np.float32)
    p = rs.rand(6, 24).astype(np.float32)
    for sym in range(4):
        bo, po = e.symmetry_apply(sym, torch.tensor(b).cuda(), torch.tensor(p).cuda())
        rb, rp = T.apply_symmetry(b, p, sym)
        assert np.array_equal(bo.cpu().numpy(), rb) and np.array_equal(po.cpu().numpy(), rp)
    with pytest.raises(DbazError):
        e.symmetry_apply(5, torch.tensor(b).cuda(), torch.tensor(p).cuda())
    e.close()


@pytest.mark.parametrize("split", ["train", "val"])
@pytest.mark.parametrize("avg", [False, True])
def test_dataset_build_matches_reference(split, avg):
    """Same seeds as the generator: np.random.seed(22) for coach's train/val split, seed(23) for
    HDFStoreDataset's df.sample; the rows sit in HBM in the reference DataFrame's order."""
    import torch
    from dotsboxesaz_amd.train_data import ReplayStore
    e = _engine(3, 3)
    rows = torch.tensor(pack_rows(G["ds_x"], G["ds_visits"], G["ds_z"], G["ds_game_idx"], G["ds_move_idx"])).cuda()
    assert rows.shape[1] == e.row_bytes
    store = ReplayStore(e)
    np.random.seed(22)
    store.add_generation(2, rows, train_split=0.9)
    np.random.seed(23)
    ds = store.dataset(train=(split == "train"), min_generation=0, pos_average=avg)
    k = "ds_%s_%s_" % (split, "avg" if avg else "raw")
    assert len(ds) == len(G[k + "value"])
    x, pi, z = e.dataset_fetch()
    assert np.array_equal(x.astype(np.float32).reshape(-1, 3, 4, 4), G[k + "features"])
    assert np.array_equal(pi, G[k + "policy"])
    assert np.array_equal(z, G[k + "value"])
    f0, p0, v0 = ds[3]                       # HDFStoreDataset.__getitem__
    assert np.array_equal(f0, G[k + "features"][3]) and np.array_equal(p0, G[k + "policy"][3])
    assert v0.shape == (1,) and v0[0] == G[k + "value"][3]
    e.close()


def test_loader_epochs_match_reference():
    """DataLoader(shuffle=True, drop_last=True) + SymmetriesGenerator of nn.py:186-216, two epochs."""
    import torch
    from dotsboxesaz_amd.train_data import ReplayStore, SymmetriesGenerator
    e = _engine(3, 3)
    rows = torch.tensor(pack_rows(G["ds_x"], G["ds_visits"], G["ds_z"])).cuda()
    store = ReplayStore(e)
    np.random.seed(22)
    store.add_generation(2, rows, train_split=0.9)
    np.random.seed(23)
    ds = store.dataset(train=True, pos_average=True)
    seed_t, seed_r, bs, epochs = (int(v) for v in G["ld_cfg"])
    torch.manual_seed(seed_t)
    random.seed(seed_r)
    loader = ds.loader(bs, shuffle=True, drop_last=True, symmetries=SymmetriesGenerator(e))
    got = []
    for _ in range(epochs):
        for boards, pi, z in loader:
            assert boards.is_cuda and boards.shape == (bs, 3, 4, 4) and pi.shape == (bs, 32) and z.shape == (bs, 1)
            got.append((boards.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()))
    assert len(got) == len(G["ld_boards"]) == epochs * (len(ds) // bs)
    for j, (b, p, z) in enumerate(got):
        assert np.array_equal(b, G["ld_boards"][j]), j
        assert np.array_equal(p, G["ld_pi"][j]), j
        assert np.array_equal(z, G["ld_z"][j]), j
    e.close()


def test_dataset_from_engine_rows_two_generations():
    """Rows produced by the engine itself, two generations in separate device buffers, where-clause,
    interleaved sampling order: raw and pos_average datasets equal the oracle on the same selection."""
    import torch
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd.self_play import _DevBuf
    from dotsboxesaz_amd.train_data import ReplayStore
    from oracle import train_ref as T
    e = Engine(3, 3, 16, mcts_num_read=20, noise=(0.8, 0.25), evaluator="formula", seed=3)
    store = ReplayStore(e)
    host = []
    np.random.seed(5)
    for gen in (0, 1, 2):
        e.selfplay_start(24, gen * 100)
        e.run()
        ptr, n, rb = e.replay_rows_dev()
        rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=torch.device("cuda", 0)).view(n, rb).clone()
        raw = rows.cpu().numpy()
        meta = raw[:, :28].copy().view(ROWMETA).ravel()
        host.append((raw[:, 28:28 + 96].copy().view(np.int16).reshape(n, 48),
                     raw[:, 124:124 + 128].copy().view(np.int32).reshape(n, 32), meta["z"].copy()))
        e.fetch_samples()
        store.add_generation(gen, rows)
    for avg in (False, True):
        for train in (True, False):
            np.random.seed(11)
            ds = store.dataset(train=train, min_generation=1, pos_average=avg)
            # the same selection on the host
            np.random.seed(11)
            chunks = [c for c in store.chunks if c["generation"] >= 1]
            cand = [(c["generation"], loc) for c in chunks for loc in (c["train_locs"] if train else c["val_locs"])]
            take = np.random.choice(len(cand), size=len(cand), replace=False)
            xs = np.stack([host[cand[t][0]][0][cand[t][1]] for t in take])
            vs = np.stack([host[cand[t][0]][1][cand[t][1]] for t in take])
            zs = np.array([host[cand[t][0]][2][cand[t][1]] for t in take])
            f, p, v = T.assemble_dataset(xs, vs, zs, avg)
            x, pi, z = e.dataset_fetch()
            assert len(ds) == len(f)
            assert np.array_equal(x, f) and np.array_equal(pi, p) and np.array_equal(z, v)
            if avg and train:
                assert len(f) < len(xs)
    e.close()


def test_dataset_errors():
    import torch
    from dotsboxesaz_amd._lib import DbazError
    e = _engine(3, 3)
    rows = torch.tensor(pack_rows(G["ds_x"], G["ds_visits"], G["ds_z"])).cuda()
    with pytest.raises(DbazError):
        e.dataset_batch([0], 0)                                  # no dataset yet
    e.dataset_begin()
    with pytest.raises(DbazError):
        e.dataset_add_rows((rows.data_ptr(), rows.shape[0], rows.shape[1] - 8))   # wrong row size
    with pytest.raises(DbazError):
        e.dataset_add_rows(rows, [0, 1, 10 ** 6])                # selection out of range
    bad = pack_rows(G["ds_x"] * 3, G["ds_visits"], G["ds_z"])   # edge planes not 0/1
    with pytest.raises(DbazError):
        e.dataset_add_rows(torch.tensor(bad).cuda())
    e.dataset_begin()
    e.dataset_add_rows(rows, [0, 1, 2, 3])
    with pytest.raises(DbazError):
        e.dataset_finish(False, order=[0, 1, 1, 2])              # not a permutation
    assert e.dataset_finish(False, order=[3, 2, 1, 0]) == 4
    x, _, _ = e.dataset_fetch()
    assert np.array_equal(x, G["ds_x"][[3, 2, 1, 0]])
    with pytest.raises(DbazError):
        e.dataset_batch([0, 4], 0)                               # batch index out of range
    with pytest.raises(DbazError):
        e.dataset_batch([0], 8)
    b, p, z = e.dataset_batch([], 0)
    assert b.shape[0] == 0
    e.dataset_begin()
    assert e.dataset_finish(True) == 0                           # empty dataset
    e.close()


def test_dataset_6x6_large_pos_average_vs_oracle():
    """~40k rows of 6x6 self-play (two key words, 2-byte aligned visit counts in the packed row):
    full comparison with the oracle, plus the size-independent properties."""
    import torch
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd.self_play import _DevBuf
    from oracle import train_ref as T
    e = Engine(6, 6, 256, mcts_num_read=12, noise=(0.8, 0.25), evaluator="uniform", seed=1)
    e.selfplay_start(512, 0)
    e.run()
    ptr, n, rb = e.replay_rows_dev()
    assert n > 30000 and rb == e.row_bytes == 720
    rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=torch.device("cuda", 0)).view(n, rb).clone()
    raw = rows.cpu().numpy()
    xs = raw[:, 28:28 + 294].copy().view(np.int16).reshape(n, 147)
    vs = raw[:, 322:322 + 392].copy().view(np.int32).reshape(n, 98)
    zs = raw[:, 25].copy().view(np.int8)
    e.dataset_begin()
    e.dataset_add_rows(rows)
    m = e.dataset_finish(True)
    x, pi, z = e.dataset_fetch()
    f, p, v = T.assemble_dataset(xs, vs, zs, True)
    assert m == len(f) < n
    assert np.array_equal(x, f) and np.array_equal(pi, p) and np.array_equal(z, v)
    # properties: rows of pi sum to 1, groups strictly ascending in the lexicographic order of x
    assert np.allclose(pi.sum(1), 1.0, atol=1e-5)
    d = np.diff(x.astype(np.int32), axis=0)
    first = (d != 0).argmax(1)
    assert np.all(d[np.arange(len(d)), first] > 0)
    # a batch under symmetry s equals the oracle transform of the gathered rows; two flips compose to identity
    idx = np.random.RandomState(0).randint(0, m, size=4096)
    for sym in range(8):
        b, pp, zz = e.dataset_batch(idx, sym)
        rb_, rp_ = T.apply_symmetry(f[idx].astype(np.float32).reshape(-1, 3, 7, 7), p[idx], sym)
        assert np.array_equal(b.cpu().numpy(), rb_) and np.array_equal(pp.cpu().numpy(), rp_)
        assert np.array_equal(zz.cpu().numpy().ravel(), v[idx])
    b1, p1, _ = e.dataset_batch(idx, 1)
    b2, p2 = e.symmetry_apply(1, b1, p1)
    b0, p0, _ = e.dataset_batch(idx, 0)
    assert torch.equal(b2, b0) and torch.equal(p2, p0)
    e.close()

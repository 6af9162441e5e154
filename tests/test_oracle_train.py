"""Oracle (oracle/train_ref.py) vs the golden vectors of the reference's training data path
(tests/golden/train.npz: SymmetriesGenerator, HDFStoreDataset's array build incl. pos_average)."""
import os

import numpy as np
import pytest

from oracle import train_ref as T

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "train.npz"))


@pytest.mark.parametrize("board", ["33", "66"])
@pytest.mark.parametrize("sym", range(8))
def test_symmetries_match_reference(board, sym):
    k = "sym%s_" % board
    b, p = T.apply_symmetry(G[k + "boards"], G[k + "pol"], sym)
    assert np.array_equal(b, G[k + "boards_out%d" % sym])
    assert np.array_equal(p, G[k + "pol_out%d" % sym])
    H, W = G[k + "boards"].shape[-2:]
    lut = T.symmetry_lut(H, W, sym)
    assert sorted(lut.tolist()) == list(range(2 * H * W))  # a permutation of the action slots
    assert np.array_equal(G[k + "pol"][:, lut], G[k + "pol_out%d" % sym])
    assert np.array_equal(G[k + "boards"][:, :2].reshape(5, -1)[:, lut].reshape(5, 2, H, W), b[:, :2])
    assert np.array_equal(G[k + "boards"][:, 2], b[:, 2])


def test_symmetry_group_structure():
    for H in (4, 7, 10):
        luts = [T.symmetry_lut(H, H, s) for s in range(8)]
        ident = np.arange(2 * H * H)
        assert np.array_equal(luts[0], ident)
        for s in (1, 2, 3, 4):   # flips and the plain transpose are involutions
            assert np.array_equal(luts[s][luts[s]], ident)
        assert len({tuple(l) for l in luts}) == 8
    # non-square boards: flips only
    T.symmetry_lut(3, 4, 3)
    with pytest.raises(ValueError):
        T.symmetry_lut(3, 4, 4)


@pytest.mark.parametrize("split", ["train", "val"])
@pytest.mark.parametrize("avg", [False, True])
def test_dataset_build_matches_reference(split, avg):
    k = "ds_%s_%s_" % (split, "avg" if avg else "raw")
    order = G[k + "order"]
    f, p, v = T.assemble_dataset(G["ds_x"][order], G["ds_visits"][order], G["ds_z"][order], avg)
    assert np.array_equal(f.astype(np.float32).reshape(-1, 3, 4, 4), G[k + "features"])
    assert np.array_equal(p, G[k + "policy"])   # float32 of the float64 (Kahan) means: bit-exact
    assert np.array_equal(v, G[k + "value"])
    if avg and split == "train":
        assert len(f) < len(order)              # the fixture does contain repeated positions

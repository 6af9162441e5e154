"""Oracle (oracle/dbaz_oracle.c) vs golden vectors of the reference's BoxesState
(dots_boxes_game.py:10-118) -- rows G1-G6 of SURVEY.md section 8a."""
import numpy as np
import pytest

from oracle import oracle as O


def _check_state(d, s, g, key, i):
    A = d.A
    assert np.array_equal(O.board_array(d, s).ravel(), g[key + "_board"][i][:A])
    assert s.to_play == g[key + "_to_play"][i]
    assert s.just_played == g[key + "_just_played"][i]
    assert [s.b2c2[0], s.b2c2[1]] == list(g[key + "_b2c2"][i])
    r = O.get_result(s)
    assert (2 if r is None else r) == g[key + "_result"][i]
    assert np.array_equal(O.valid_moves(d, s), g[key + "_valid"][i])
    assert np.array_equal(O.features(d, s).ravel(), g[key + "_features"][i])
    assert [int(s.hash_bits[w]) for w in range(4)] == [int(x) for x in g[key + "_hash_words"][i]]
    assert s.hash_b2c2 == g[key + "_hash_b2c2"][i]


def test_playouts_bit_exact(golden_rules):
    g = golden_rules
    for (r, c) in g["boards"]:
        d = O.dims(int(r), int(c))
        gi = 0
        while "b%dx%d_g%d_moves" % (r, c, gi) in g:
            key = "b%dx%d_g%d" % (r, c, gi)
            s = O.new_state(d)
            _check_state(d, s, g, key, 0)
            for i, m in enumerate(g[key + "_moves"]):
                closed = O.play_(d, s, int(m))
                assert len(closed) == g[key + "_closed_n"][i]
                exp = [int(x) for x in g[key + "_closed_lc"][i] if x >= 0]
                assert [x for lc in closed for x in lc] == exp
                _check_state(d, s, g, key, i + 1)
            gi += 1
        assert gi >= 1


def test_illegal_moves(golden_rules):
    g = golden_rules
    for (r, c) in g["boards"]:
        d = O.dims(int(r), int(c))
        s = O.new_state(d)
        O.play_(d, s, 0)
        got = []
        for m in (0, int(c), d.A - 1):
            before = bytes(s.board)
            try:
                O.play_(d, s, m)
                got.append(0)
            except ValueError:
                got.append(1)
                assert bytes(s.board) == before
        assert got == list(g["b%dx%d_illegal" % (r, c)])
    with pytest.raises(ValueError):
        O.play_(d, s, d.A)
    with pytest.raises(ValueError):
        O.play_(d, s, -1)


def test_reference_csv_fixture(golden_boards):
    """The 34 sequences of the reference's test/test_boards.csv replay legally and
    reach the states the reference reaches (test/nn_unittests.py:5-20)."""
    g = golden_boards
    d = O.dims(3, 3)
    assert len(g["ids"]) == 34
    for i in g["ids"]:
        k = "id%d_" % i
        s = O.state_from_moves(d, g[k + "moves"])
        assert np.array_equal(O.board_array(d, s).ravel(), g[k + "board"])
        assert np.array_equal(O.features(d, s).ravel(), g[k + "features"])
        r = O.get_result(s)
        assert [s.to_play, s.just_played, s.b2c2[0], s.b2c2[1], 2 if r is None else r] == list(g[k + "meta"])
        vm = O.valid_moves(d, s)
        assert all(vm[m] for m in g[k + "next_moves"])


@pytest.mark.parametrize("n", [1, 7, 8, 9, 24, 32, 98, 127, 128, 129, 200, 256])
def test_numpy_pairwise_sum(n):
    rng = np.random.RandomState(n)
    for _ in range(20):
        a = (rng.rand(n).astype(np.float32) * rng.choice([1e-3, 1.0, 1e3])).astype(np.float32)
        assert O.np_sum_f32(a) == a.sum()
        b = rng.rand(n) * 1e3
        assert O.np_sum_f64(b) == b.sum()

"""HIP rules kernels (through the C ABI) vs the oracle and the reference's golden
vectors -- rows G1-G6.  Bit-exact."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    from dotsboxesaz_amd.engine import Engine
    cache = {}

    def get(r, c):
        if (r, c) not in cache:
            cache[(r, c)] = Engine(r, c, 4, mcts_num_read=8)
        return cache[(r, c)]
    yield get
    for e in cache.values():
        e.close()


def edges_of(words):
    return [int(w) for w in words]


def test_golden_playouts(engines, golden_rules):
    g = golden_rules
    for (r, c) in g["boards"]:
        r, c = int(r), int(c)
        e = engines(r, c)
        keys = []
        gi = 0
        while "b%dx%d_g%d_moves" % (r, c, gi) in g:
            keys.append("b%dx%d_g%d" % (r, c, gi))
            gi += 1
        n = len(keys)
        st = e.rules_init(n)
        maxlen = max(len(g[k + "_moves"]) for k in keys)

        def check(i_ply, alive):
            vm = e.rules_valid_moves(st)
            res = e.rules_result(st)
            ft = e.rules_features(st)
            for s, k in enumerate(keys):
                if not alive[s]:
                    continue
                assert edges_of(st["edges"][s]) == edges_of(g[k + "_hash_words"][i_ply]), (k, i_ply)
                assert list(st["b2c2"][s]) == list(g[k + "_b2c2"][i_ply])
                assert st["to_play"][s] == g[k + "_to_play"][i_ply]
                assert st["just_played"][s] == g[k + "_just_played"][i_ply]
                assert res[s] == g[k + "_result"][i_ply]
                assert np.array_equal(vm[s], g[k + "_valid"][i_ply])
                assert np.array_equal(ft[s].ravel(), g[k + "_features"][i_ply])

        check(0, [True] * n)
        for i in range(maxlen):
            alive = [i < len(g[k + "_moves"]) for k in keys]
            # finished games replay an already-played edge: must be reported illegal, state untouched
            moves = np.array([g[k + "_moves"][i] if alive[s] else g[k + "_moves"][0] for s, k in enumerate(keys)], np.int32)
            before = {f: st[f].copy() for f in st}
            nc, lc = e.rules_play_status(st, moves)
            for s, k in enumerate(keys):
                if alive[s]:
                    assert nc[s] == g[k + "_closed_n"][i], (k, i)
                    assert np.array_equal(lc[s], g[k + "_closed_lc"][i])
                else:
                    assert nc[s] == -1
                    for f in st:
                        assert np.array_equal(st[f][s], before[f][s])
            check(i + 1, alive)


def test_illegal_move_raises_value_error(engines, golden_rules):
    g = golden_rules
    for (r, c) in g["boards"]:
        e = engines(int(r), int(c))
        st = e.rules_init(1)
        e.rules_play(st, [0])
        got = []
        for m in (0, int(c), e.A - 1):
            try:
                e.rules_play(st, [m])
                got.append(0)
            except ValueError:
                got.append(1)
        assert got == list(g["b%dx%d_illegal" % (r, c)])
        for m in (-1, e.A, 10 ** 6):
            with pytest.raises(ValueError):
                e.rules_play(st, [m])


def test_reference_csv_positions(engines, golden_boards):
    g = golden_boards
    e = engines(3, 3)
    ids = [int(i) for i in g["ids"]]
    st = e.rules_init(len(ids))
    maxlen = max(len(g["id%d_moves" % i]) for i in ids)
    for ply in range(maxlen):
        # shorter sequences are padded with an illegal (already played) move
        mv = np.array([g["id%d_moves" % i][ply] if ply < len(g["id%d_moves" % i]) else g["id%d_moves" % i][0]
                       for i in ids], np.int32)
        e.rules_play_status(st, mv)
    ft = e.rules_features(st)
    res = e.rules_result(st)
    for s, i in enumerate(ids):
        k = "id%d_" % i
        assert np.array_equal(ft[s].ravel(), g[k + "features"])
        assert [st["to_play"][s], st["just_played"][s], st["b2c2"][s][0], st["b2c2"][s][1], res[s]] == list(g[k + "meta"])


# (10, 10) and (15, 7): the largest boards the library accepts (DBAZ_MAX_A: 2 (rows + 1)(cols + 1) <= 256)
@pytest.mark.parametrize("rows,cols,n", [(3, 3, 2048), (6, 6, 4096), (9, 9, 1024), (2, 5, 512), (7, 1, 300), (10, 10, 256), (15, 7, 256)])
def test_random_playouts_vs_oracle(rows, cols, n):
    """Thousands of concurrent seeded playouts, every ply compared with the C oracle."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(rows, cols, 4, mcts_num_read=8)
    d = O.dims(rows, cols)
    rng = np.random.RandomState(rows * 100 + cols)
    st = e.rules_init(n)
    ost = [O.new_state(d) for _ in range(n)]
    check_idx = rng.choice(n, size=min(n, 192), replace=False)
    for ply in range(e.E + 2):
        vm = e.rules_valid_moves(st)
        # 1 in 16 states tries an illegal move (played edge / sentinel / out of range)
        moves = np.zeros(n, np.int32)
        for s in range(n):
            legal = np.nonzero(vm[s])[0]
            if len(legal) == 0 or rng.randint(16) == 0:
                moves[s] = rng.choice([rng.randint(-3, e.A + 3), int(np.nonzero(~vm[s])[0][0])])
            else:
                moves[s] = legal[rng.randint(len(legal))]
        nc, _ = e.rules_play_status(st, moves)
        for s in range(n):
            try:
                cl = O.play_(d, ost[s], int(moves[s]))
                assert nc[s] == len(cl)
            except ValueError:
                assert nc[s] == -1
        res = e.rules_result(st)
        ft = e.rules_features(st)
        vm2 = e.rules_valid_moves(st)
        for s in range(n):
            o = ost[s]
            assert [int(o.hash_bits[w]) for w in range(4)] == edges_of(st["edges"][s])
            assert (o.b2c2[0], o.b2c2[1], o.to_play, o.just_played) == (st["b2c2"][s][0], st["b2c2"][s][1], st["to_play"][s], st["just_played"][s])
            r = O.get_result(o)
            assert res[s] == (2 if r is None else r)
        for s in check_idx:
            assert np.array_equal(ft[s], O.features(d, ost[s]))
            assert np.array_equal(vm2[s], O.valid_moves(d, ost[s]))
    # finish the boards with legal moves only (the illegal attempts above cost plies)
    for _ in range(e.E):
        vm = e.rules_valid_moves(st)
        if not vm.any():
            break
        moves = np.array([int(np.argmax(vm[s])) if vm[s].any() else 0 for s in range(n)], np.int32)
        e.rules_play_status(st, moves)
    # every board is full at the end: encode -> play-out -> terminal (size-independent property)
    assert not e.rules_valid_moves(st).any()
    assert (e.rules_result(st) != 2).all()
    assert (st["b2c2"].astype(int).sum(1) == 0).all()  # closed boxes sum to B
    e.close()


def test_empty_batch(engines):
    e = engines(3, 3)
    st = e.rules_init(0)
    assert e.rules_valid_moves(st).shape == (0, e.A)
    assert e.rules_result(st).shape == (0,)

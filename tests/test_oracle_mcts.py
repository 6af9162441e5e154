"""Oracle sequential MCTS vs golden vectors of the reference's mcts.py
(UCT_search with max_pending_evals=1, init_mcts_tree) -- rows M1-M9."""
import numpy as np
import pytest

from oracle import oracle as O
from conftest import load_golden

_G = load_golden("mcts.npz")
CASES = [str(c) for c in _G["cases"]]


def replay_case(g, name, make_tree, check):
    """Shared driver: walks a golden script and calls check(tree, key) after each search."""
    rows, cols, kind, c0, c1 = g[name + "_cfg"]
    d = O.dims(int(rows), int(cols))
    start = O.state_from_moves(d, g[name + "_start"])
    tree = make_tree(d, start, int(kind))
    for si, (op, a, b, c) in enumerate(g[name + "_script"]):
        key = "%s_s%d_" % (name, si)
        if op == 0:
            noise = g[key + "noise"] if b > 0 else None
            vis = tree.search(int(a), cpuct=(c0, c1), dirichlet=(b, c), noise=noise)
            check(tree, key, vis)
        else:
            tree.advance(int(a), bool(b))
    return tree


class _OracleTree:
    def __init__(self, d, start, kind):
        self.t = O.Tree(d, start)
        self.ev = O.Evaluator(kind)

    def search(self, n, cpuct, dirichlet, noise):
        return self.t.search(n, self.ev, cpuct=cpuct, dirichlet=dirichlet, noise=noise)

    def advance(self, move, reuse):
        self.t.advance(move, reuse)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    g = _G

    def check(tree, key, vis):
        pri, tv, nv, pc = tree.t.root_arrays()
        assert np.array_equal(vis, g[key + "visits"]), key
        assert np.array_equal(nv, g[key + "visits"]), key
        assert np.array_equal(tv.view(np.uint32), g[key + "total_value"].view(np.uint32)), key
        assert np.array_equal(pri.view(np.uint64), g[key + "priors"].view(np.uint64)), key
        assert np.array_equal(pc, g[key + "changed"]), key
        md, ts, tc, q = tree.t.stats()
        assert [md, ts, tc] == list(g[key + "stats_i"]), key
        assert np.float32(q).view(np.uint32) == g[key + "q"].view(np.uint32), key
        rtv, rnv = tree.t.root_slot()
        assert rnv == g[key + "root_nv"] and np.float32(rtv).view(np.uint32) == g[key + "root_tv"].view(np.uint32)

    replay_case(g, name, _OracleTree, check)


def test_formula_evaluator_is_hash_pure():
    """NN-cache contract (utils/proxies.py:20,35-38): the evaluation is a function of
    get_hash() only -- two move orders reaching one position evaluate identically."""
    d = O.dims(3, 3)
    a = O.state_from_moves(d, [0, 5, 9, 20])
    b = O.state_from_moves(d, [9, 20, 0, 5])
    pa, va = O.eval_formula(d, a)
    pb, vb = O.eval_formula(d, b)
    assert np.array_equal(pa, pb) and va == vb
    c = O.state_from_moves(d, [0, 5, 9, 21])
    pc, _ = O.eval_formula(d, c)
    assert not np.array_equal(pa, pc)


def test_illegal_advance_raises():
    d = O.dims(3, 3)
    t = O.Tree(d, O.new_state(d))
    t.search(5, O.Evaluator(0))
    t.advance(0)
    with pytest.raises(ValueError):
        t.advance(0)


# ---- max_pending_evals = K > 1: the oracle's wave search vs the reference's UCT_search under a
# once-suspending evaluator (tests/golden/pending.npz, written by gen_golden.py --only pending)
_P = load_golden("pending.npz")
PCASES = [str(c) for c in _P["cases"]]


@pytest.mark.parametrize("name", PCASES)
def test_oracle_pending_waves_match_reference(name):
    g = _P
    rows, cols, kind, c0, c1, K = g[name + "_cfg"]
    d = O.dims(int(rows), int(cols))
    t = O.Tree(d, O.state_from_moves(d, g[name + "_start"]))
    ev = O.Evaluator(int(kind))
    for si, (op, a, b, c) in enumerate(g[name + "_script"]):
        key = "%s_s%d_" % (name, si)
        if op == 0:
            vis = t.search(int(a), ev, cpuct=(c0, c1), dirichlet=(b, c), noise=g[key + "noise"] if b > 0 else None,
                           max_pending=int(K))
            pri, tv, nv, pc = t.root_arrays()
            assert np.array_equal(vis, g[key + "visits"]), key
            assert np.array_equal(tv.view(np.uint32), g[key + "total_value"].view(np.uint32)), key
            assert np.array_equal(pri.view(np.uint64), g[key + "priors"].view(np.uint64)), key
            assert np.array_equal(pc, g[key + "changed"]), key
            md, ts, tc, q = t.stats()
            assert [md, ts, tc] == list(g[key + "stats_i"]), key
            assert np.float32(q).view(np.uint32) == g[key + "q"].view(np.uint32), key
            rtv, rnv = t.root_slot()
            assert rnv == g[key + "root_nv"] and np.float32(rtv).view(np.uint32) == g[key + "root_tv"].view(np.uint32)
        else:
            t.advance(int(a), bool(b))


# The reference's own search test (test/mcts_tests.py:80-124): on every row of test/test_boards.csv, after 800 sequential
# reads (max_pending_evals = 1, no noise, cpuct (1.25, 19652)) the most visited move must be one of the row's `next_moves`.
# Its mock evaluator -- uniform priors + the mean of 100 random playouts -- depends on numpy's global RNG (26 of 34
# rows pass, the rest need the trained net); with the deterministic evaluators of this repository the rows below satisfy it.
CSV_BEST_MOVE_ROWS = {
    0: [1, 2, 3, 4, 5, 6, 7, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -16, -17, -18, -19, -20, -21, -24],
    1: [1, 2, 4, 5, -1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11, -12, -13, -14, -15, -16, -17, -18, -19, -20, -21, -22, -24],
}


def csv_searches(golden_boards, kind):
    """(id, start moves, acceptable moves, oracle visit counts after 800 reads) per non-terminal CSV row"""
    g = golden_boards
    d = O.dims(3, 3)
    out = []
    for i in [int(x) for x in g["ids"]]:
        k = "id%d_" % i
        st = O.state_from_moves(d, g[k + "moves"])
        assert O.get_result(st) is None
        vis = O.Tree(d, st).search(800, O.Evaluator(kind), cpuct=(1.25, 19652.0), dirichlet=(0.0, 0.0), noise=None)
        out.append((i, [int(m) for m in g[k + "moves"]], {int(m) for m in g[k + "next_moves"]}, vis))
    return out


@pytest.mark.parametrize("kind", [0, 1])
def test_reference_csv_best_moves(golden_boards, kind):
    ok = [i for i, _, nxt, vis in csv_searches(golden_boards, kind) if int(np.argmax(vis)) in nxt]
    assert ok == CSV_BEST_MOVE_ROWS[kind]
    if kind == 1:  # uniform priors, value 0 off the terminals: every tactical row but one is found by search alone
        assert set(ok) >= {1, 2, 4, 5} | set(range(-21, 0))

"""UCT_search with several pending evaluations on one tree (mcts.py:228-239; players.AZPlayer):
k_select_multi / k_expand_backup_multi through the C ABI.
  * one simulation in flight: the kernels reproduce the sequential search -- the 35 golden scripts -- bit for bit;
  * K > 1 with the reference's bookkeeping (virtual_visits = 0): bit-exact against tests/golden/pending.npz, i.e. against the
    reference's own UCT_search(..., max_pending_evals=K) under an evaluator that suspends once per call (its asyncio loop
    then runs the searches in synchronous waves of K; other evaluator schedules -- the batching proxy's 50 ms timer --
    interleave differently and are not reproduced);
  * K > 1 with counted virtual visits (the default of the AZPlayer mirror; not a reference mode: parity unpinned): the
    bookkeeping identities of mcts.py, determinism, and the wall-clock cut-off."""
import time

import numpy as np
import pytest

from oracle import oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu
_G = load_golden("mcts.npz")
CASES = [str(c) for c in _G["cases"]]


@pytest.mark.parametrize("name", CASES)
def test_one_pending_equals_reference_golden(name):
    from dotsboxesaz_amd.engine import Engine
    g = _G
    rows, cols, kind, c0, c1 = g[name + "_cfg"]
    e = Engine(int(rows), int(cols), 2, mcts_num_read=800, evaluator="uniform" if kind == 1 else "formula", max_pending_evals=8)
    e.set_pending(1)
    st = list(g[name + "_start"])
    e.set_positions([st, st])
    for si, (op, a, b, c) in enumerate(g[name + "_script"]):
        key = "%s_s%d_" % (name, si)
        if op == 0:
            e.set_search_params((c0, c1), (b, c))
            e.search(int(a), None if b <= 0 else np.stack([g[key + "noise"]] * 2))
            r = e.roots()
            for s in (0, 1):
                assert np.array_equal(r["visits"][s], g[key + "visits"]), key
                assert np.array_equal(r["total_value"][s].view(np.uint32), g[key + "total_value"].view(np.uint32)), key
                assert np.array_equal(r["priors"][s].view(np.uint64), g[key + "priors"].view(np.uint64)), key
                assert list(r["stats"][s]) == list(g[key + "stats_i"]), key
                assert r["q"][s].view(np.uint32) == g[key + "q"].view(np.uint32), key
                assert r["root_nv"][s] == g[key + "root_nv"]
        else:
            e.advance(int(a), bool(b))
    e.close()


def _starts(d, n_slots, rng):
    starts = []
    for s in range(n_slots):
        st, mv = O.new_state(d), []
        for _ in range(rng.randint(0, max(1, d.A // 3))):
            legal = np.nonzero(O.valid_moves(d, st))[0]
            m = int(legal[rng.randint(len(legal))])
            tmp = st.copy()
            O.play_(d, tmp, m)
            if O.get_result(tmp) is not None:
                break
            st = tmp
            mv.append(m)
        starts.append(mv)
    return starts


@pytest.mark.parametrize("rows,cols,n_slots,sims,K", [(3, 3, 32, 200, 8), (6, 6, 16, 500, 64), (6, 6, 4, 333, 5), (9, 9, 6, 300, 64)])
def test_pending_waves_bookkeeping(rows, cols, n_slots, sims, K):
    from dotsboxesaz_amd.engine import Engine
    d = O.dims(rows, cols)
    starts = _starts(d, n_slots, np.random.RandomState(K + sims))

    def run():
        e = Engine(rows, cols, n_slots, mcts_num_read=sims, evaluator="formula", max_pending_evals=K)
        e.set_positions(starts)
        e.search(sims)
        r, c = e.roots(), e.counters()
        e.advance(np.argmax(r["visits"], axis=1).astype(np.int32), True)
        e.search(sims)
        r2, c2 = e.roots(), e.counters()
        e.close()
        return r, c, r2, c2

    r, c, r2, c2 = run()
    # every read backs up through the root; the search that expanded the root visits no child (mcts.py:121-126,207-208)
    assert (r["root_nv"] == sims + 1).all() and (r["visits"].sum(1) == sims).all()
    assert c["expansions"] == n_slots * (sims + 1) and c["error_slots"] == 0
    assert np.isfinite(r["total_value"]).all() and np.isfinite(r["q"]).all()
    # a wave's simulations spread: at K >= 8 the most visited child of a fresh root holds well under all of the visits
    assert (r["visits"].max(1) < sims).all()
    # after the re-root the kept subtree arrives with its visits (tree_size); its root was expanded
    assert (r2["stats"][:, 1] > 0).all()
    # (reads of one wave that ended at the kept child while its evaluation was pending went no deeper: up to K of them)
    below = r2["visits"].sum(1) - sims
    assert (below <= r2["stats"][:, 1] - 1).all() and (below >= r2["stats"][:, 1] - K).all() and (r2["root_nv"] == sims).all()
    assert c2["expansions"] == n_slots * (2 * sims + 1)
    # evaluations + shared leaves (selected again while pending) + terminal leaves account for every read
    assert c2["nn_evals"] + c2["cache_hits"] + c2["terminal_leaves"] == c2["expansions"]
    # deterministic: a second handle reproduces every number
    s, cs, s2, cs2 = run()
    for k in ("visits", "total_value", "priors", "stats"):
        assert np.array_equal(r[k], s[k]) and np.array_equal(r2[k], s2[k]), k
    assert cs2 == c2 or all(cs2[k] == c2[k] for k in ("expansions", "nn_evals", "cache_hits", "terminal_leaves", "sum_path"))


def test_pending_with_network_and_time_limit():
    """AZPlayer's call: UCT_search(node, int(1e12), nn, cpuct, max_async_searches=64, (0, 0), time_limit): one 6x6 tree,
    64 reads in flight, random-init ResNetZero on the device; stops on the clock, leaves a usable tree."""
    import torch
    from oracle import nn_ref
    from dotsboxesaz_amd.engine import Engine
    torch.manual_seed(0)
    m = nn_ref.ResNetZeroRef(6, 6, 64, 4)
    nn_ref.randomize_bn(m, 2)
    e = Engine(6, 6, 1, mcts_num_read=800, evaluator="resnet", nn_precision=1, nodes_per_slot=400000, max_pending_evals=64)
    e.load_state_dict(m.state_dict(), "resnet", 64, 4, 16, 8)
    e.set_positions(None)
    t0 = time.time()
    e.search_timed(0.5, num_reads=2 ** 31 - 1)
    dt = time.time() - t0
    r, c = e.roots(), e.counters()
    assert 0.45 < dt < 3.0
    reads = int(r["root_nv"][0]) - 1
    assert reads > 2000 and r["visits"][0].sum() == reads and c["expansions"] == reads + 1
    assert c["nn_evals"] + c["cache_hits"] + c["terminal_leaves"] == c["expansions"]
    print("single 6x6 tree, 64 pending, ResNetZero 4x64: %.0f reads/s" % (reads / dt))
    mv = int(np.argmax(r["visits"][0]))
    e.advance([mv], True)
    e.search(100)
    r2 = e.roots()
    below = r2["visits"][0].sum() - 100
    assert r2["root_nv"][0] == 100 and r2["stats"][0, 1] - 64 <= below <= r2["stats"][0, 1] - 1
    e.close()


def test_az_player_mirror_time_limited_move():
    """players.AZPlayer request through the mirrors: create_root_uct_node(state, nn=model) + UCT_search(node, int(1e12),
    ..., max_async_searches, (0, 0), time_limit) on the device."""
    import torch
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd.game import BoxesState
    from dotsboxesaz_amd.players import AZPlayer
    BoxesState.init_static_fields(((3, 3),))
    params = dnn.resnet_params(3, 3, 32, 2)
    params["nn"]["model_class"] = dnn.ResNetZero
    params["self_play"] = {"mcts": {"mcts_cpuct": (1.25, 19652), "max_async_searches": 16}}
    torch.manual_seed(1)
    player = AZPlayer(params, time_limit=0.3)
    s = BoxesState()
    for m in (0, 5, 17):
        s.play_(m)
    t0 = time.time()
    move, policy, rate = player.get_move(s)
    dt = time.time() - t0
    assert move is not None and s.get_valid_moves()[move] and policy[move] == policy.max() > 0
    assert policy.sum() > 500 and 0.25 < dt < 5.0 and rate > 1000
    assert (policy[~np.asarray(s.get_valid_moves(), bool)] == 0).all()


# ---- K > 1 against the reference itself: tests/golden/pending.npz holds UCT_search(..., max_pending_evals=K) of the
# reference under an evaluator that suspends once per call, which makes its asyncio loop run the searches in synchronous
# waves of K (first wave min(K, A)): K selections, then K expand + backups.  With virtual_visits = 0 the device kernels keep
# the reference's bookkeeping and must agree bit for bit -- duplicates of a pending leaf, terminal leaves, partial last waves,
# tree reuse and noise included.
_P = load_golden("pending.npz")
PCASES = [str(c) for c in _P["cases"]]


@pytest.mark.parametrize("name", PCASES)
def test_pending_waves_equal_reference_golden(name):
    from dotsboxesaz_amd.engine import Engine
    g = _P
    rows, cols, kind, c0, c1, K = g[name + "_cfg"]
    K = int(K)
    e = Engine(int(rows), int(cols), 2, mcts_num_read=800, evaluator="uniform" if kind == 1 else "formula", max_pending_evals=K)
    e.set_pending(K, virtual_visits=False)
    st = list(g[name + "_start"])
    e.set_positions([st, st])
    shared = 0
    for si, (op, a, b, c) in enumerate(g[name + "_script"]):
        key = "%s_s%d_" % (name, si)
        if op == 0:
            e.set_search_params((c0, c1), (b, c))
            e.search(int(a), None if b <= 0 else np.stack([g[key + "noise"]] * 2))
            r = e.roots()
            for s in (0, 1):
                assert np.array_equal(r["visits"][s], g[key + "visits"]), key
                assert np.array_equal(r["total_value"][s].view(np.uint32), g[key + "total_value"].view(np.uint32)), key
                assert np.array_equal(r["priors"][s].view(np.uint64), g[key + "priors"].view(np.uint64)), key
                assert np.array_equal(r["changed"][s], g[key + "changed"]), key
                assert list(r["stats"][s]) == list(g[key + "stats_i"]), key
                assert r["q"][s].view(np.uint32) == g[key + "q"].view(np.uint32), key
                assert r["root_nv"][s] == g[key + "root_nv"] and r["root_tv"][s].view(np.uint32) == g[key + "root_tv"].view(np.uint32)
        else:
            e.advance(int(a), bool(b))
    shared = e.counters()["cache_hits"]
    if K >= 8 and "n7" not in name:
        assert shared > 0  # the reference's pending simulations do pile onto the same leaves; they share one evaluation here
    e.close()


@pytest.mark.parametrize("rows,cols,n_slots,sims,K", [(3, 3, 96, 150, 8), (6, 6, 48, 300, 64), (4, 2, 40, 90, 3), (9, 9, 12, 250, 64),
                                                       (6, 6, 32, 200, 16)])
def test_pending_waves_many_slots_vs_oracle(rows, cols, n_slots, sims, K):
    """Every slot searches a different position with its own read count; three searches with tree reuse, noise on the second:
    root arrays against the oracle's wave search (itself pinned to the reference by pending.npz), slot by slot."""
    from dotsboxesaz_amd.engine import Engine
    d = O.dims(rows, cols)
    rng = np.random.RandomState(K * 1000 + sims)
    starts = _starts(d, n_slots, rng)
    e = Engine(rows, cols, n_slots, mcts_num_read=sims, evaluator="formula", max_pending_evals=K)
    e.set_pending(K, virtual_visits=False)
    e.set_positions(starts)
    trees = [O.Tree(d, O.state_from_moves(d, mv)) for mv in starts]
    ev = O.Evaluator(0)
    reads = rng.randint(1, sims + 1, size=n_slots).astype(np.int32)
    for rnd, (alpha, coeff) in enumerate([(0.0, 0.0), (0.8, 0.25), (0.0, 0.0)]):
        noise = rng.dirichlet(np.full(d.A, 0.8), size=n_slots) if alpha > 0 else None
        e.set_search_params((1.25, 19652), (alpha, coeff))
        e.search(reads, noise)
        r = e.roots()
        moves = np.zeros(n_slots, np.int32)
        for s in range(n_slots):
            vis = trees[s].search(int(reads[s]), ev, dirichlet=(alpha, coeff), noise=None if noise is None else noise[s],
                                  max_pending=K)
            pri, tv, nv, pc = trees[s].root_arrays()
            assert np.array_equal(r["visits"][s], vis), (rnd, s)
            assert np.array_equal(r["total_value"][s].view(np.uint32), tv.view(np.uint32)), (rnd, s)
            assert np.array_equal(r["priors"][s].view(np.uint64), pri.view(np.uint64)), (rnd, s)
            md, ts, tc, q = trees[s].stats()
            assert list(r["stats"][s]) == [md, ts, tc] and r["q"][s].view(np.uint32) == np.float32(q).view(np.uint32)
            moves[s] = int(np.argmax(vis)) if s % 2 == 0 else int(rng.choice(np.nonzero(O.valid_moves(d, trees[s].state))[0]))
        for s in range(n_slots):
            tmp = trees[s].state
            O.play_(d, tmp, int(moves[s]))
            if O.get_result(tmp) is not None:
                moves[s] = -1
            else:
                trees[s].advance(int(moves[s]), True)
        e.advance(moves, True)
    e.close()


@pytest.mark.parametrize("K", [1, 8, 64])
def test_pending_waves_with_the_network_vs_oracle_fed_by_it(K):
    """Searches of several 6x6 trees with K simulations in flight and the ResNetZero evaluator (the reference's bookkeeping,
    virtual_visits = 0), against the oracle's wave search whose evaluator asks the same engine for (p, v) of one position at a
    time: root arrays and statistics bit-identical, across a re-root with tree reuse."""
    import torch
    from oracle import nn_ref
    from dotsboxesaz_amd.engine import Engine
    rows = cols = 6
    d = O.dims(rows, cols)
    rng = np.random.RandomState(40 + K)
    n_slots, sims = 6, 160
    starts = _starts(d, n_slots, rng)
    torch.manual_seed(K)
    m = nn_ref.ResNetZeroRef(rows, cols, 64, 6)
    nn_ref.randomize_bn(m, 3)
    e = Engine(rows, cols, n_slots, mcts_num_read=sims, evaluator="resnet", nn_precision=1, max_pending_evals=max(K, 2))
    e.load_state_dict(m.state_dict(), "resnet", 64, 6, 16, 8)
    e.set_pending(K, virtual_visits=False)
    e.set_positions(starts)
    memo = {}

    def hip_net(dd, st):
        x = O.features(dd, st)
        key = x.tobytes()
        if key not in memo:
            pv = e.predict(x.astype(np.float32).reshape(1, 3, rows + 1, cols + 1))
            memo[key] = (pv[0][0].copy(), pv[1][0].copy())
        return memo[key]

    ev = O.Evaluator(hip_net)
    trees = [O.Tree(d, O.state_from_moves(d, mv)) for mv in starts]
    reads = rng.randint(sims // 2, sims + 1, size=n_slots).astype(np.int32)
    for rnd in range(2):
        e.set_search_params((1.25, 19652), (0.0, 0.0))
        e.search(reads, None)
        r = e.roots()
        moves = np.zeros(n_slots, np.int32)
        for s in range(n_slots):
            vis = trees[s].search(int(reads[s]), ev, dirichlet=(0.0, 0.0), noise=None, max_pending=K)
            pri, tv, nv, pc = trees[s].root_arrays()
            assert np.array_equal(r["visits"][s], vis), (rnd, s)
            assert np.array_equal(r["total_value"][s].view(np.uint32), tv.view(np.uint32)), (rnd, s)
            assert np.array_equal(r["priors"][s].view(np.uint64), pri.view(np.uint64)), (rnd, s)
            md, ts, tc, q = trees[s].stats()
            assert list(r["stats"][s]) == [md, ts, tc] and r["q"][s].view(np.uint32) == np.float32(q).view(np.uint32)
            moves[s] = int(np.argmax(vis))
        for s in range(n_slots):
            tmp = trees[s].state
            O.play_(d, tmp, int(moves[s]))
            if O.get_result(tmp) is not None:
                moves[s] = -1
            else:
                trees[s].advance(int(moves[s]), True)
        e.advance(moves, True)
    e.close()


def _fact_reads(n_valid, sims):
    """n_searches = min(4 * factorial(nb_valid_moves), mcts_num_read), self_play.py:64-65"""
    f = 4.0
    for i in range(2, n_valid + 1):
        f *= i
        if f > sims:
            break
    return int(f) if f < sims else sims


@pytest.mark.parametrize("rows,cols,n_slots,n_games,sims,K,reuse", [(3, 3, 8, 20, 60, 8, True), (6, 6, 6, 6, 100, 8, True),
                                                                   (3, 3, 4, 8, 40, 64, False), (2, 3, 5, 10, 30, 3, True)])
def test_selfplay_driver_in_waves_vs_oracle_wave_search(rows, cols, n_slots, n_games, sims, K, reuse):
    """dbaz_config.selfplay_pending (self_play.py:27-30: every self-play search runs with max_async_searches simulations of the
    tree in flight): the DRIVER -- dbaz_run with slot refill, 4*n! rule, temperature sampling, re-rooting, row emission -- steps
    every game in waves of K through k_select_multi / k_expand_backup_multi with the reference's bookkeeping.  Each finished game is
    replayed move by move by the oracle's wave search (ob_uct_search_pending, pinned to the reference's own
    UCT_search(..., max_pending_evals=K) by tests/golden/pending.npz), teacher-forced with the moves the device sampled: visit
    vectors, pi, q, TreeStats, features and z of every row must be bit-identical."""
    from dotsboxesaz_amd.engine import Engine
    d = O.dims(rows, cols)
    e = Engine(rows, cols, n_slots, mcts_num_read=sims, noise=(0.0, 0.0), reuse_tree=reuse, evaluator="formula", seed=3,
               max_pending_evals=K, selfplay_pending=True)
    e.selfplay_start(n_games, 0)
    e.run()
    c = e.counters()
    assert c["games_finished"] == n_games and c["error_slots"] == 0
    got = e.fetch_samples()
    e.close()
    ev = O.Evaluator(0)
    n_exp = 0
    for gidx in range(n_games):
        sel = np.nonzero(got["game_idx"] == gidx)[0]
        assert list(got["move_idx"][sel]) == list(range(len(sel)))
        t = O.Tree(d, O.new_state(d))
        for r in sel:
            st = t.state
            nv = int(O.valid_moves(d, st).sum())
            reads = _fact_reads(nv, sims)
            vis = t.search(reads, ev, max_pending=K)
            md, ts, tc, q = t.stats()
            n_exp += reads
            assert np.array_equal(got["visits"][r], vis), (gidx, r)
            assert np.array_equal(got["x"][r], O.features(d, st).ravel()), (gidx, r)
            assert (int(got["max_deepness"][r]), int(got["tree_size"][r]), int(got["terminal_count"][r])) == (md, ts, tc), (gidx, r)
            assert np.float32(got["q_value"][r]).view(np.uint32) == np.float32(q).view(np.uint32), (gidx, r)
            assert got["player"][r] == st.to_play
            s = vis.sum()
            assert np.array_equal(got["pi"][r], vis.astype(np.float64) / (s if s else 1.0))
            t.advance(int(got["played"][r]), reuse)
        assert t.is_terminal
        # z: +z_T for rows whose to_play is the terminal state's just_played, -z_T otherwise (self_play.py:105-112)
        term = t.state
        zt = O.get_result(term)
        for r in sel:
            exp = zt if got["player"][r] == term.just_played else -zt
            assert int(got["z"][r]) == exp, (gidx, r)
    assert n_exp > 0


def test_engine_kwargs_honour_max_async_searches_only_on_request():
    from dotsboxesaz_amd.self_play import engine_kwargs_from_params
    params = {"self_play": {"reuse_mcts_tree": True, "noise": [0.8, 0.25],
                            "mcts": {"mcts_num_read": 20, "mcts_cpuct": [1.25, 19652], "temperature": {0: 1.0}, "max_async_searches": 64}}}
    assert "max_pending_evals" not in engine_kwargs_from_params(params)
    kw = engine_kwargs_from_params(params, async_searches=True)
    assert kw["max_pending_evals"] == 64 and kw["selfplay_pending"] is True

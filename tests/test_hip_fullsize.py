"""Full BASELINE sizes: size-independent properties over ALL games (rules invariants, tree bookkeeping identities,
dataset-row consistency -- the oracle cannot follow 8192 games x 800 sims), and, with the noise off and the hash-formula
evaluator, the rows of a sample of the games replayed search by search with the oracle, bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def check_rows(got, rows, cols, n_games):
    d = O.dims(rows, cols)
    assert sorted(set(got["game_idx"])) == list(range(n_games))
    assert np.allclose(got["pi"].sum(1), 1.0, atol=1e-12)
    assert (got["visits"][np.arange(len(got["played"])), got["played"]] > 0).all()
    # replay a sample of games with the oracle rules: legality, features, player, z
    for gi in list(range(0, n_games, max(1, n_games // 64)))[:64]:
        r = np.nonzero(got["game_idx"] == gi)[0]
        assert np.array_equal(got["move_idx"][r], np.arange(len(r)))
        s = O.new_state(d)
        for i in r:
            assert np.array_equal(got["x"][i], O.features(d, s).ravel())
            assert got["player"][i] == s.to_play
            O.play_(d, s, int(got["played"][i]))
        res = O.get_result(s)
        assert res in (0, 1)
        assert np.array_equal(got["z"][r], np.where(got["player"][r] == s.just_played, res, -res))


def test_config1_3x3_4096_games_100_sims():
    """BASELINE configs[1]: 3x3, 4096 concurrent games, 100 sims/move, complete games."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(3, 3, 4096, mcts_num_read=100, noise=(0.8, 0.25), evaluator="formula", seed=11)
    e.selfplay_start(4096, 0)
    e.run()
    c = e.counters()
    assert c["games_finished"] == 4096 and c["error_slots"] == 0 and c["active_slots"] == 0
    assert c["pool_resets"] == 0                      # the default pool never has to give a reused subtree up here
    got = e.fetch_samples()
    check_rows(got, 3, 3, 4096)
    # every move consumed min(4*n_valid!, 100) reads (+1 root expansion on the first move of a game)
    assert c["expansions"] >= len(got["z"]) * 4
    e.close()


def test_config1_full_size_rows_vs_oracle_searches():
    """configs[1] at full size with the noise off (numpy's Dirichlet stream cannot be matched on the device): every 16th
    of the 4 096 games is replayed by the oracle, teacher-forced with the device's moves -- 100-read searches with tree
    reuse from a fully loaded engine; visit counts, pi, q and TreeStats of every row bit-identical."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(3, 3, 4096, mcts_num_read=100, noise=(0.0, 0.0), evaluator="formula", seed=12)
    e.selfplay_start(4096, 0)
    e.run()
    c = e.counters()
    assert c["games_finished"] == 4096 and c["error_slots"] == 0 and c["pool_resets"] == 0
    got = e.fetch_samples()
    d = O.dims(3, 3)
    pp = O.selfplay_params(100, noise=(0.0, 0.0), reuse_tree=True)
    ev = O.Evaluator(0)
    for gi in range(0, 4096, 16):
        r = np.nonzero(got["game_idx"] == gi)[0]
        ref = O.play_game(d, pp, ev, forced_moves=got["played"][r])
        assert ref["n_rows"] == len(r)
        assert np.array_equal(ref["visits"], got["visits"][r])
        assert np.array_equal(ref["pi"].view(np.uint64), got["pi"][r].view(np.uint64))
        assert np.array_equal(ref["q_value"].view(np.uint32), got["q_value"][r].view(np.uint32))
        assert np.array_equal(ref["tree_size"], got["tree_size"][r])
        assert np.array_equal(ref["terminal_count"], got["terminal_count"][r])
        assert np.array_equal(ref["max_deepness"], got["max_deepness"][r].astype(np.int32))
        assert np.array_equal(ref["z"], got["z"][r].astype(np.int64))
    e.close()


def test_config2_full_size_rows_vs_oracle_searches():
    """configs[2] at full size -- 8 192 concurrent 6x6 games, 800 reads per move, tree reuse, complete games -- with the
    hash-formula evaluator and the noise off: 16 of the games are replayed by the oracle, teacher-forced with the device's
    moves; every row's visit counts, pi, q and TreeStats bit-identical."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(6, 6, 8192, mcts_num_read=800, noise=(0.0, 0.0), evaluator="formula", seed=13)
    e.selfplay_start(8192, 0)
    e.run()
    c = e.counters()
    assert c["games_finished"] == 8192 and c["error_slots"] == 0 and c["pool_resets"] == 0
    got = e.fetch_samples()
    d = O.dims(6, 6)
    pp = O.selfplay_params(800, noise=(0.0, 0.0), reuse_tree=True)
    ev = O.Evaluator(0)
    for gi in range(0, 8192, 512):
        r = np.nonzero(got["game_idx"] == gi)[0]
        ref = O.play_game(d, pp, ev, forced_moves=got["played"][r])
        assert ref["n_rows"] == len(r)
        assert np.array_equal(ref["visits"], got["visits"][r])
        assert np.array_equal(ref["pi"].view(np.uint64), got["pi"][r].view(np.uint64))
        assert np.array_equal(ref["q_value"].view(np.uint32), got["q_value"][r].view(np.uint32))
        assert np.array_equal(ref["tree_size"], got["tree_size"][r])
        assert np.array_equal(ref["terminal_count"], got["terminal_count"][r])
        assert np.array_equal(ref["max_deepness"], got["max_deepness"][r].astype(np.int32))
        assert np.array_equal(ref["z"], got["z"][r].astype(np.int64))
    e.close()


def test_config2_6x6_8192_games_800_sims_steps():
    """BASELINE configs[2] at full size for a bounded number of steps: after K steps every slot
    has performed exactly K (+1 root expansion) searches and the root bookkeeping identities of
    mcts.py hold: sum(child visits) == root visits - 1, W finite, priors a distribution."""
    from dotsboxesaz_amd.engine import Engine
    K = 60
    e = Engine(6, 6, 8192, mcts_num_read=800, noise=(0.8, 0.25), evaluator="formula", seed=5)
    e.selfplay_start(1 << 30, 0)
    e.step(K)
    c = e.counters()
    assert c["error_slots"] == 0 and c["expansions"] == 8192 * K
    r = e.roots()
    assert (r["root_nv"] == K).all()                # every _search backs up through the root (mcts.py:121-126)
    assert (r["visits"].sum(1) == K - 1).all()      # the search that expanded the root visits no child
    assert np.isfinite(r["total_value"]).all() and np.isfinite(r["q"]).all()
    st = e.root_states()
    valid = e.rules_valid_moves(st)
    assert (r["visits"][~valid] == 0).all()
    assert (r["priors"][~valid] == 0).all() and np.allclose(r["priors"].sum(1), 1.0, atol=0.26)  # 0.75 + masked noise
    assert c["pool_high_water"] <= K + 1
    e.close()


def test_resnet_full_batch_network_consistency():
    """8192-sample batches through the fused trunk: identical to small batches bit for bit."""
    import torch
    from oracle import nn_ref
    from dotsboxesaz_amd.engine import Engine
    torch.manual_seed(0)
    m = nn_ref.ResNetZeroRef(6, 6, 64, 3)
    nn_ref.randomize_bn(m, 2)
    e = Engine(6, 6, 8192, evaluator="resnet", nn_precision=1)
    e.load_state_dict(m.state_dict(), "resnet", 64, 3, 16, 8)
    rng = np.random.RandomState(0)
    X = rng.randint(0, 2, size=(8192, 3, 7, 7)).astype(np.float32)
    p, v = e.predict(X)
    for lo in (0, 4093, 8188):
        p2, v2 = e.predict(X[lo:lo + 4])
        assert np.array_equal(p2, p[lo:lo + 4]) and np.array_equal(v2, v[lo:lo + 4])
    pr, vr = nn_ref.predict_sync(m, X[:64])
    assert np.abs(p[:64] - pr).max() < 1e-4 and np.abs(v[:64] - vr).max() < 1e-4
    e.close()


# ---------------------------------------------------------------- BASELINE configs[4], per-GPU share
def test_config4_9x9_2048_games_1600_sims_steps():
    """9x9 (A = 200, 3264-byte nodes, 16 020-node pools), 2048 concurrent games, 1600 sims/move -- one GPU's share
    of configs[4] -- for a bounded number of steps, first searches staggered so that re-rooting with tree reuse
    happens inside the window.  Bookkeeping identities of mcts.py / self_play.py per slot:
      still in the first search:   sum(child visits) == root visits - 1   (the expanding search visits no child)
      after a re-root (reuse):     sum(child visits) == tree_size - 1 + root visits   (mcts.py:163-180: the kept
                                   child arrives with its own visits; its first visit expanded it)"""
    from dotsboxesaz_amd.engine import Engine
    K, n = 150, 2048
    e = Engine(9, 9, n, mcts_num_read=1600, noise=(0.8, 0.25), evaluator="formula", seed=5)
    first = (np.arange(n) * 7) % 100 + 1
    first[::4] = 0  # a quarter of the slots keep the full first search
    e.selfplay_stagger(first)
    e.selfplay_start(1 << 30, 0)
    e.step(K)
    c = e.counters()
    assert c["error_slots"] == 0
    r = e.roots()
    moved = r["stats"][:, 1] > 0  # tree_size > 0: re-rooted with a kept subtree
    assert moved.sum() > n // 2 and (~moved).sum() >= n // 4
    # a slot skips the step in which the driver pass starts its next search
    assert n * K - 4 * n <= c["expansions"] <= n * K
    vs = r["visits"].sum(1)
    assert (vs[~moved] == r["root_nv"][~moved] - 1).all()
    assert (vs[moved] == r["stats"][moved, 1] - 1 + r["root_nv"][moved]).all()
    assert (r["root_nv"][~moved] == K).all()
    assert np.isfinite(r["total_value"]).all() and np.isfinite(r["q"]).all()
    st = e.root_states()
    valid = e.rules_valid_moves(st)
    assert (r["visits"][~valid] == 0).all() and (r["priors"][~valid] == 0).all()
    assert (r["stats"][:, 0] >= 0).all() and (r["stats"][:, 0] <= e.E + 1).all()          # max_deepness
    assert c["pool_high_water"] <= K + 2 and c["moves_played"] >= moved.sum()
    e.close()


def test_config4_full_size_rows_vs_oracle_searches():
    """The per-GPU share of configs[4] at full size -- 2 048 concurrent 9x9 games, 1 600 reads per move, 16 020-node pools,
    complete games -- hash-formula evaluator, noise off: 4 of the games replayed by the oracle, teacher-forced with the
    device's moves; every row's visit counts, pi, q and TreeStats bit-identical."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(9, 9, 2048, mcts_num_read=1600, noise=(0.0, 0.0), evaluator="formula", seed=14)
    e.selfplay_start(2048, 0)
    e.run()
    c = e.counters()
    assert c["games_finished"] == 2048 and c["error_slots"] == 0 and c["pool_resets"] == 0
    got = e.fetch_samples()
    d = O.dims(9, 9)
    pp = O.selfplay_params(1600, noise=(0.0, 0.0), reuse_tree=True)
    ev = O.Evaluator(0)
    for gi in range(0, 2048, 512):
        r = np.nonzero(got["game_idx"] == gi)[0]
        ref = O.play_game(d, pp, ev, forced_moves=got["played"][r])
        assert ref["n_rows"] == len(r)
        assert np.array_equal(ref["visits"], got["visits"][r])
        assert np.array_equal(ref["pi"].view(np.uint64), got["pi"][r].view(np.uint64))
        assert np.array_equal(ref["q_value"].view(np.uint32), got["q_value"][r].view(np.uint32))
        assert np.array_equal(ref["tree_size"], got["tree_size"][r])
        assert np.array_equal(ref["terminal_count"], got["terminal_count"][r])
        assert np.array_equal(ref["max_deepness"], got["max_deepness"][r].astype(np.int32))
        assert np.array_equal(ref["z"], got["z"][r].astype(np.int64))
    e.close()


def test_config4_9x9_complete_games_vs_oracle_rules():
    """A few complete 9x9 games at 1600 sims/move (176+ plies each): every row replayed with the oracle rules
    (legality, features, player, z), read budget rule, pool high-water and depth bounds."""
    from dotsboxesaz_amd.engine import Engine
    n_games = 6
    e = Engine(9, 9, n_games, mcts_num_read=1600, noise=(0.8, 0.25), evaluator="formula", seed=21)
    e.selfplay_start(n_games, 0)
    e.run()
    c = e.counters()
    assert c["games_finished"] == n_games and c["error_slots"] == 0 and c["active_slots"] == 0 and c["pool_resets"] == 0
    got = e.fetch_samples()
    check_rows(got, 9, 9, n_games)
    cap = 10 * (1600 + 2)
    assert 1600 < c["pool_high_water"] <= cap
    # deepest node below a root cannot lie beyond the end of the game
    remaining = e.E - got["move_idx"].astype(np.int32)
    assert (got["max_deepness"] >= 0).all() and (got["max_deepness"] <= remaining + 1).all()
    assert (got["tree_size"][got["move_idx"] == 0] == 0).all() and (got["tree_size"] <= 1601 * got["move_idx"].astype(np.int64)).all()  # carried visits accumulate over the moves
    per_game = c["expansions"] / n_games
    assert 150 * 1600 < per_game <= (e.E + 1) * 1601
    e.close()


@pytest.mark.parametrize("cut", [False, True])
def test_config4_9x9_network_steps(cut):
    """The fused trunk on 10x10 images (2 samples per workgroup, 13 position tiles) inside the self-play step at 2048
    slots: every expansion is a network evaluation, a cache hit or a terminal leaf."""
    import torch
    from oracle import nn_ref
    from dotsboxesaz_amd.engine import Engine
    torch.manual_seed(0)
    m = nn_ref.ResNetZeroRef(9, 9, 64, 4)
    nn_ref.randomize_bn(m, 2)
    K, n = 12, 2048
    # eval_round = -1: every leaf is evaluated in the step that selected it
    e = Engine(9, 9, n, mcts_num_read=1600, noise=(0.8, 0.25), evaluator="resnet", nn_precision=1, seed=3, eval_round=0 if cut else -1)
    e.load_state_dict(m.state_dict(), "resnet", 64, 4, 16, 8)
    e.selfplay_fastforward((np.arange(n) * 37) % 120)
    e.selfplay_start(1 << 30, 0)
    e.step(K)
    c = e.counters()
    assert c["error_slots"] == 0
    assert c["nn_evals"] + c["cache_hits"] + c["terminal_leaves"] == c["expansions"] and c["nn_evals"] > n * K // 2
    r = e.roots()
    if cut:
        # full rounds only (nn.hip cut_n): a slot whose leaf falls behind the step's last full round of 512 (when at most 256
        # would be left over) completes that search a step later -- never more than one search per step, none lost
        assert n * K * 3 // 4 < c["expansions"] < n * K
        assert (r["root_nv"] <= K).all() and (r["root_nv"] >= K // 2).all() and (r["visits"].sum(1) == r["root_nv"] - 1).all()
    else:
        assert c["expansions"] == n * K
        assert (r["root_nv"] == K).all() and (r["visits"].sum(1) == K - 1).all()
    assert np.isfinite(r["total_value"]).all() and np.allclose(r["priors"].sum(1), 1.0, atol=0.26)
    e.close()

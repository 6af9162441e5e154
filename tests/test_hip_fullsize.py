"""Full BASELINE sizes through size-independent properties (the oracle cannot follow 8192 games x
800 sims): rules invariants, tree bookkeeping identities and dataset-row consistency."""
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
    got = e.fetch_samples()
    check_rows(got, 3, 3, 4096)
    # every move consumed min(4*n_valid!, 100) reads (+1 root expansion on the first move of a game)
    assert c["expansions"] >= len(got["z"]) * 4
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

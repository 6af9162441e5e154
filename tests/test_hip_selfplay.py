"""HIP self-play driver (device-side SelfPlay.play_game + get_datasets rows) vs the
reference's golden games and the oracle -- rows D1-D3, B1."""
import numpy as np
import pytest

from oracle import oracle as O
from conftest import load_golden
from test_oracle_selfplay import golden_games

pytestmark = pytest.mark.gpu

_G = load_golden("selfplay.npz")
FORMULA_CASES = [str(c) for c in _G["cases"] if str(c) != "sp33_resnet"]


def compare_rows(got, rows_slice, g, name):
    r = rows_slice
    assert np.array_equal(got["move"], g[name + "_move"][r])
    assert np.array_equal(got["player"], g[name + "_player"][r])
    assert np.array_equal(got["x"], g[name + "_x"][r])
    assert np.array_equal(got["pi"].view(np.uint64), g[name + "_pi"][r].view(np.uint64))
    assert np.array_equal(got["z"].astype(np.int64), g[name + "_z"][r])
    st = np.stack([got["max_deepness"].astype(np.int32), got["tree_size"], got["terminal_count"]], axis=1)
    assert np.array_equal(st, g[name + "_stats"][r])
    assert np.array_equal(got["q_value"].view(np.uint32), g[name + "_q"][r].view(np.uint32))
    assert np.array_equal(got["move_idx"], g[name + "_index"][r, 2])
    assert np.array_equal(got["game_idx"], g[name + "_index"][r, 1])


# tt="force": the per-game transposition table (B1, utils/proxies.py:35-43) switched on for the formula evaluators, so
# the reference's golden games and the oracle replays also pin its hit path (see test_hip_mcts.TT_MODES)
@pytest.mark.parametrize("tt", [True, "force"])
@pytest.mark.parametrize("name", FORMULA_CASES)
@pytest.mark.parametrize("n_slots", [1, 4])
def test_golden_games_teacher_forced(name, n_slots, tt):
    """The reference's sampled moves and Dirichlet vectors are injected; every row of
    get_datasets must match bit for bit (n_slots=1 plays the games one after the other in
    one slot, n_slots=4 concurrently)."""
    from dotsboxesaz_amd.engine import Engine
    g = _G
    rows, cols, sims, a, c, reuse, n_games, _seed = g[name + "_cfg"]
    temp = {int(k): float(v) for k, v in g[name + "_temp"]}
    e = Engine(int(rows), int(cols), n_slots, mcts_num_read=int(sims), noise=(a, c), temperature=temp,
               reuse_tree=bool(reuse), evaluator="uniform" if name == "sp33_uniform" else "formula", transposition_cache=tt)
    games = golden_games(g, name)
    for gi, gg in enumerate(games):
        e.selfplay_script(gi, gg["moves"], gg["noise"])
    e.selfplay_start(len(games), 0)
    e.run()
    cnt = e.counters()
    assert cnt["games_finished"] == len(games) and cnt["error_slots"] == 0
    assert (cnt["cache_hits"] > 0) == (tt == "force")  # whole games with tree reuse: transpositions always occur
    got = e.fetch_samples()
    all_rows = np.concatenate([gg["rows"] for gg in games])
    assert np.array_equal(got["played"], np.concatenate([gg["moves"] for gg in games]))
    compare_rows(got, all_rows, g, name)
    e.close()


@pytest.mark.parametrize("tt", [True, "force"])
@pytest.mark.parametrize("rows,cols,n_slots,n_games,sims,reuse", [(3, 3, 64, 200, 40, True), (3, 3, 32, 70, 30, False),
                                                                  (6, 6, 48, 48, 60, True), (2, 3, 16, 40, 50, True),
                                                                  (10, 10, 6, 6, 24, True)])  # the last: largest square board
def test_device_sampled_games_vs_oracle(rows, cols, n_slots, n_games, sims, reuse, tt):
    """Production path: moves sampled on the device (Philox), slots refilled as games end.
    The oracle replays each game teacher-forced with the device's moves; all rows must be
    bit-identical (noise off: numpy's Dirichlet stream cannot be matched on the device)."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(rows, cols, n_slots, mcts_num_read=sims, noise=(0.0, 0.0), reuse_tree=reuse, evaluator="formula",
               seed=1234, transposition_cache=tt)
    e.selfplay_start(n_games, 100)
    e.run()
    cnt = e.counters()
    assert cnt["games_finished"] == n_games and cnt["active_slots"] == 0 and cnt["error_slots"] == 0
    assert (cnt["cache_hits"] > 0) == (tt == "force")
    got = e.fetch_samples()
    assert sorted(set(got["game_idx"])) == list(range(100, 100 + n_games))
    d = O.dims(rows, cols)
    pp = O.selfplay_params(sims, noise=(0.0, 0.0), reuse_tree=reuse)
    ev = O.Evaluator(0)
    total_search = 0
    for gi in range(100, 100 + n_games):
        r = np.nonzero(got["game_idx"] == gi)[0]
        ref = O.play_game(d, pp, ev, forced_moves=got["played"][r])
        total_search += ref["n_search"]
        assert ref["n_rows"] == len(r)
        assert np.array_equal(ref["move"], got["move"][r])
        assert np.array_equal(ref["player"], got["player"][r])
        assert np.array_equal(ref["x"], got["x"][r])
        assert np.array_equal(ref["visits"], got["visits"][r])
        assert np.array_equal(ref["pi"].view(np.uint64), got["pi"][r].view(np.uint64))
        assert np.array_equal(ref["z"], got["z"][r].astype(np.int64))
        assert np.array_equal(ref["q_value"].view(np.uint32), got["q_value"][r].view(np.uint32))
        assert np.array_equal(ref["tree_size"], got["tree_size"][r])
        assert np.array_equal(ref["terminal_count"], got["terminal_count"][r])
        assert np.array_equal(ref["max_deepness"], got["max_deepness"][r].astype(np.int32))
        # legality + scoring: replaying the moves reaches a finished game with the recorded winner
        s = O.new_state(d)
        for m in got["played"][r]:
            O.play_(d, s, int(m))
        assert O.get_result(s) in (0, 1)
    assert cnt["expansions"] == total_search  # node-expansion counter == number of _search calls
    e.close()


def test_device_noise_is_a_valid_dirichlet_mix():
    """With device-drawn noise the root priors stay a probability vector over the valid moves:
    0.75*probs + 0.25*noise, noise >= 0, masked (mcts.py:219-226)."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(6, 6, 64, mcts_num_read=30, noise=(0.8, 0.25), evaluator="formula", seed=7)
    e.set_positions(None)
    e.search(30)
    r = e.roots()
    f = Engine(6, 6, 64, mcts_num_read=30, noise=(0.0, 0.0), evaluator="formula")
    f.set_positions(None)
    f.search(30)
    base = f.roots()["priors"]
    noise = (r["priors"] - 0.75 * base) / 0.25
    st = e.root_states()
    valid = e.rules_valid_moves(st)
    assert (noise[~valid] == 0).all() and (noise >= -1e-12).all()
    # the Dirichlet sample sums to 1 over ALL slots, so the valid part sums to <= 1 and is not degenerate
    s = noise.sum(1)
    assert (s <= 1 + 1e-9).all() and (s > 0.5).all()
    assert np.abs(noise[0] - noise[1]).max() > 1e-3  # slots draw different vectors
    e.close()
    f.close()


def test_resnet_selfplay_end_to_end():
    """BASELINE config 1 shape (3x3, 25 sims, random-init ResNetZero) fully on the device:
    legal finished games, consistent rows, and the first root's (p, v) equals predict()."""
    import torch
    from oracle import nn_ref
    from dotsboxesaz_amd.engine import Engine
    torch.manual_seed(0)
    m = nn_ref.ResNetZeroRef(3, 3, 64, 4)
    nn_ref.randomize_bn(m, 3)
    e = Engine(3, 3, 16, mcts_num_read=25, noise=(0.8, 0.25), evaluator="resnet", seed=3)
    e.load_state_dict(m.state_dict(), "resnet", 64, 4, 16, 8)
    e.selfplay_start(32, 0)
    e.run()
    cnt = e.counters()
    assert cnt["games_finished"] == 32 and cnt["error_slots"] == 0
    got = e.fetch_samples()
    d = O.dims(3, 3)
    for gi in range(32):
        r = np.nonzero(got["game_idx"] == gi)[0]
        s = O.new_state(d)
        for k, i in enumerate(r):
            assert np.array_equal(got["x"][i], O.features(d, s).ravel())
            assert got["player"][i] == s.to_play and got["move_idx"][i] == k
            assert abs(got["pi"][i].sum() - 1) < 1e-12 and got["visits"][i][got["played"][i]] > 0
            O.play_(d, s, int(got["played"][i]))
        res = O.get_result(s)
        assert res in (0, 1)
        zexp = np.where(got["player"][r] == s.just_played, res, -res)
        assert np.array_equal(got["z"][r], zexp)
    e.close()


@pytest.mark.parametrize("rows,cols,n_slots,n_games,sims,blocks,precision", [(3, 3, 16, 32, 25, 4, 0), (3, 3, 16, 32, 25, 4, 1),
                                                                            (6, 6, 8, 3, 40, 20, 1), (6, 6, 2048, 2048, 30, 20, 1)])
def test_resnet_selfplay_rows_vs_oracle_fed_by_the_hip_network(rows, cols, n_slots, n_games, sims, blocks, precision):
    """Self-play with the network evaluator, bit for bit: the games the engine played (device-sampled moves, leaves of all
    games in one batch per step, transposition table on) are replayed by the oracle's sequential search, whose evaluator
    asks the same engine for (p, v) of one position at a time (predict_sync) -- a sample's result does not depend on its
    batch (utils/proxies.py:35-43), so every row must come out identical: visits, pi, q, TreeStats, z.  The 2 048-slot case
    runs whole rounds of the two-cout-tile kernel, the remainder bodies and the full-rounds-only cut (4 of its games are
    replayed); predict_sync of one position runs the smallest one-cout-tile body."""
    import torch
    from oracle import nn_ref
    from dotsboxesaz_amd.engine import Engine
    torch.manual_seed(rows * 10 + blocks)
    m = nn_ref.ResNetZeroRef(rows, cols, 64, blocks)
    nn_ref.randomize_bn(m, 3)
    e = Engine(rows, cols, n_slots, mcts_num_read=sims, noise=(0.0, 0.0), evaluator="resnet", seed=5, nn_precision=precision)
    e.load_state_dict(m.state_dict(), "resnet", 64, blocks, 16, 8)
    e.selfplay_start(n_games, 0)
    e.run()
    cnt = e.counters()
    assert cnt["games_finished"] == n_games and cnt["error_slots"] == 0 and cnt["cache_hits"] > 0
    got = e.fetch_samples()
    d = O.dims(rows, cols)
    memo = {}

    def hip_net(dd, st):
        x = O.features(dd, st)
        key = x.tobytes()
        if key not in memo:
            pv = e.predict(x.astype(np.float32).reshape(1, 3, rows + 1, cols + 1))
            memo[key] = (pv[0][0].copy(), pv[1][0].copy())
        return memo[key]

    ev = O.Evaluator(hip_net)
    pp = O.selfplay_params(sims, noise=(0.0, 0.0), reuse_tree=True)
    for gi in range(0, n_games, 1 if n_games <= 32 else n_games // 4):
        r = np.nonzero(got["game_idx"] == gi)[0]
        ref = O.play_game(d, pp, ev, forced_moves=got["played"][r])
        assert ref["n_rows"] == len(r)
        assert np.array_equal(ref["visits"], got["visits"][r]), gi
        assert np.array_equal(ref["pi"].view(np.uint64), got["pi"][r].view(np.uint64))
        assert np.array_equal(ref["q_value"].view(np.uint32), got["q_value"][r].view(np.uint32))
        assert np.array_equal(ref["tree_size"], got["tree_size"][r])
        assert np.array_equal(ref["terminal_count"], got["terminal_count"][r])
        assert np.array_equal(ref["max_deepness"], got["max_deepness"][r].astype(np.int32))
        assert np.array_equal(ref["z"], got["z"][r].astype(np.int64))
    e.close()


def test_output_buffer_backpressure():
    """A tiny finished-row buffer: finished games wait (PH_EMIT), run() drains and resumes;
    nothing is lost or duplicated."""
    from dotsboxesaz_amd.engine import Engine
    e = Engine(3, 3, 8, mcts_num_read=20, evaluator="formula", seed=5, max_out_rows=30)
    e.selfplay_start(24, 0)
    e.run()
    got = e.fetch_samples()
    assert e.counters()["games_finished"] == 24
    assert sorted(set(got["game_idx"])) == list(range(24))
    for gi in range(24):
        r = np.nonzero(got["game_idx"] == gi)[0]
        assert np.array_equal(got["move_idx"][r], np.arange(len(r)))
    e.close()


def test_replay_rows_stay_in_hbm_and_match_fetch():
    """dbaz_replay_rows_dev: the packed rows the RCCL all-gather ships (RowMeta | x | visits),
    wrapped zero-copy as a torch tensor through __cuda_array_interface__, equal fetch_samples()."""
    import torch
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd.self_play import _DevBuf
    e = Engine(3, 3, 8, mcts_num_read=20, evaluator="formula", seed=9)
    e.selfplay_start(8, 0)
    e.run()
    ptr, n, rb = e.replay_rows_dev()
    assert n > 0 and rb == (28 + 48 * 2 + 32 * 4 + 7) // 8 * 8
    rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=torch.device("cuda", 0)).view(n, rb).cpu().numpy()
    got = e.fetch_samples()
    assert len(got["z"]) == n
    meta = rows[:, :28].copy()
    game = meta[:, 0:4].copy().view(np.int32).ravel()
    move_idx = meta[:, 4:6].copy().view(np.int16).ravel()
    order = np.lexsort((move_idx, game))
    x = rows[:, 28:28 + 96].copy().view(np.int16).reshape(n, 48)[order]
    vis = rows[:, 28 + 96:28 + 96 + 128].copy().view(np.int32).reshape(n, 32)[order]
    z = meta[:, 25].copy().view(np.int8)[order]
    assert np.array_equal(x, got["x"]) and np.array_equal(vis, got["visits"]) and np.array_equal(z, got["z"])
    assert np.array_equal(game[order], got["game_idx"]) and np.array_equal(move_idx[order], got["move_idx"])
    e.close()


# ---------------------------------------------------------------- match play
@pytest.mark.parametrize("n_slots", [1, 8])
def test_match_play_golden(n_slots):
    """Two evaluators in one engine (model 0 = hash formula, model 1 = uniform), the model of the
    player to move at the root serves the move's search, seats swapped on odd games; rows equal the
    reference's compute_elo game loop bit for bit (teacher-forced moves)."""
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd.self_play import match_winners
    from test_oracle_selfplay import match_games
    g = load_golden("match.npz")
    for name in [str(c) for c in g["cases"]]:
        rows, cols, sims, n_games, _seed = [int(x) for x in g[name + "_cfg"]]
        e = Engine(rows, cols, n_slots, mcts_num_read=sims, noise=(0.0, 0.0), reuse_tree=False, evaluator="formula",
                   evaluator2="uniform", match_play=True)
        games = match_games(g, name)
        for gi, gg in enumerate(games):
            e.selfplay_script(gi, gg["moves"])
        e.selfplay_start(n_games, 0)
        e.run()
        got = e.fetch_samples()
        r = np.concatenate([gg["rows"] for gg in games])
        assert np.array_equal(got["played"], np.concatenate([gg["moves"] for gg in games]))
        assert np.array_equal(got["move"], g[name + "_move"][r])
        assert np.array_equal(got["player"], g[name + "_player"][r])
        assert np.array_equal(got["pi"].view(np.uint64), g[name + "_pi"][r].view(np.uint64))
        assert np.array_equal(got["z"].astype(np.int64), g[name + "_z"][r])
        assert np.array_equal(got["q_value"].view(np.uint32), g[name + "_q"][r].view(np.uint32))
        st = np.stack([got["max_deepness"].astype(np.int32), got["tree_size"], got["terminal_count"]], axis=1)
        assert np.array_equal(st, g[name + "_stats"][r])
        # winner bookkeeping of compute_elo (self_play.py:336-338)
        n0, n1 = match_winners(got, (7, 9))
        exp0 = exp1 = 0
        for gi, gg in enumerate(games):
            rr = gg["rows"]
            win = rr[g[name + "_z"][rr] == 1]
            if len(win):
                model = int(g[name + "_player"][win[0]]) ^ (gi & 1)
                exp0 += model == 0
                exp1 += model == 1
        assert (n0, n1) == (exp0, exp1)
        e.close()


def test_compute_elo_two_networks():
    """compute_elo mirror with two different ResNetZero nets on the MFMA path."""
    import torch
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd.self_play import compute_elo
    torch.manual_seed(0)
    pa = dnn.resnet_params(3, 3, 32, 2)
    pb = dnn.resnet_params(3, 3, 16, 1)
    for p in (pa, pb):
        p["self_play"] = {"reuse_mcts_tree": True, "noise": [0.8, 0.25],
                          "mcts": {"mcts_num_read": 100, "mcts_cpuct": [1.25, 19652], "temperature": {0: 1.0, 12: 0.02}}}
    elo_params = {"n_games": 24, "self_play_override": {"reuse_mcts_tree": False, "noise": [0.0, 0.0],
                                                       "mcts": {"mcts_num_read": 30}}}
    e0, e1, wins1 = compute_elo(elo_params, [pa, pb], [0, 0], (1000.0, 1000.0), nn_classes=[dnn.ResNetZero, dnn.ResNetZero],
                                rows=3, cols=3, n_slots=8)
    assert abs((e0 - 1000.0) + (e1 - 1000.0)) < 1e-9  # zero-sum update
    assert np.isnan(wins1) or 0.0 <= wins1 <= 1.0


def test_baseline_config0_reference_game_with_recorded_network_outputs():
    """BASELINE configs[0]: the reference's CPU self-play game (3x3, 25 sims/move, ResNetZero,
    noise (0.8, 0.25), tree reuse, np.random.seed(0)).  The device tree kernels replay it through the
    external-evaluator path, fed the (p, v) the reference's network returned for every leaf (recorded
    by position hash) plus its sampled moves and Dirichlet vectors: visit counts, pi, q and TreeStats
    of every move must equal the reference's get_datasets() rows bit for bit."""
    from dotsboxesaz_amd.engine import Engine
    g = _G
    name = "sp33_resnet"
    table = {}
    for k, p, v in zip(g[name + "_evalkeys"], g[name + "_evalp"], g[name + "_evalv"]):
        table[tuple(int(x) for x in k)] = (p, v)

    def evaluate(x):
        P, V = [], []
        for f in x:
            bits = 0
            for i in np.nonzero(f[:2].ravel())[0]:
                bits |= 1 << int(i)
            key = tuple((bits >> (64 * w)) & (2 ** 64 - 1) for w in range(4)) + ((int(f[2, 0, 0]) + 512) & (2 ** 64 - 1),)
            p, v = table[key]
            P.append(p)
            V.append(v)
        return np.stack(P), np.array(V).reshape(-1)

    e = Engine(3, 3, 1, mcts_num_read=25, noise=(0.8, 0.25), reuse_tree=True, evaluator="external")
    e.set_positions(None)
    moves, noise = g[name + "_drawn_moves"], g[name + "_drawn_noise"]
    assert len(moves) == len(g[name + "_z"])
    for ply in range(len(moves)):
        e.set_search_params((1.25, 19652), (0.8, 0.25))
        e.search_external(evaluate, num_reads=None, noise=noise[ply][None])
        r = e.roots()
        vis = r["visits"][0]
        pi = vis / (vis.sum() or 1.0)
        assert np.array_equal(pi.view(np.uint64), g[name + "_pi"][ply].view(np.uint64)), ply
        assert list(r["stats"][0]) == list(g[name + "_stats"][ply]), ply
        assert r["q"][0].view(np.uint32) == g[name + "_q"][ply].view(np.uint32), ply
        st = e.root_states()
        assert st["to_play"][0] == g[name + "_player"][ply]
        assert np.array_equal(e.rules_features(st)[0].ravel(), g[name + "_x"][ply])
        e.advance([int(moves[ply])], True)
    st = e.root_states()
    assert st["result"][0] in (0, 1)
    winner = st["just_played"][0]
    z = np.where(g[name + "_player"] == winner, st["result"][0], -st["result"][0])
    assert np.array_equal(z, g[name + "_z"])
    e.close()


# ---------------------------------------------------------------- transposition cache (B1)
@pytest.mark.parametrize("rows,cols,sims,precision", [(3, 3, 60, 0), (6, 6, 120, 1), (9, 9, 40, 1), (2, 5, 80, 1)])
def test_transposition_cache_is_transparent(rows, cols, sims, precision):
    """Per-game transposition table (utils/proxies.py:35-43 semantics): the same seeded self-play with the cache on
    and off produces identical rows; every hit replaces exactly one network evaluation."""
    import torch
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd import nn as dnn
    torch.manual_seed(3)
    model = dnn.ResNetZero(dnn.resnet_params(rows, cols, 32, 2, 4, 8))
    out = []
    for cache in (True, False):
        e = Engine(rows, cols, 24, mcts_num_read=sims, noise=(0.8, 0.25), reuse_tree=True, evaluator="resnet", seed=5,
                   nn_precision=precision, transposition_cache=cache)
        e.load_state_dict(model.state_dict(), "resnet", **model.shape)
        e.selfplay_start(24, 0)
        e.run()
        c = e.counters()
        out.append((e.fetch_samples(), c))
        e.close()
    (a, ca), (b, cb) = out
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert ca["expansions"] == cb["expansions"] and ca["terminal_leaves"] == cb["terminal_leaves"]
    assert cb["cache_hits"] == 0 and ca["cache_hits"] > 0.01 * ca["expansions"]
    assert ca["nn_evals"] + ca["cache_hits"] == cb["nn_evals"]


def test_rows_do_not_depend_on_slot_scheduling():
    """A game's rows are a function of (seed, game_idx) only: 96 games played on 24 slots (refills; games drift apart,
    the driver kernel re-roots some slot next to almost every network launch) equal the same 96 games played on 96
    slots in lockstep."""
    import torch
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd import nn as dnn
    torch.manual_seed(4)
    model = dnn.ResNetZero(dnn.resnet_params(3, 3, 32, 2, 4, 8))
    out = []
    for n_slots in (24, 96):
        e = Engine(3, 3, n_slots, mcts_num_read=48, noise=(0.8, 0.25), reuse_tree=True, evaluator="resnet", seed=11)
        e.load_state_dict(model.state_dict(), "resnet", **model.shape)
        e.selfplay_start(96, 0)
        e.run()
        c = e.counters()
        assert c["games_finished"] == 96 and c["error_slots"] == 0
        out.append(e.fetch_samples())
        e.close()
    a, b = out
    assert len(a["z"]) == len(b["z"])
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("rnd", [(16, 15), (16, 6), (7, 6)])
def test_full_rounds_only_changes_nothing_but_the_schedule(rnd):
    """Full rounds only (nn.hip cut_n): when a step's evaluation list would leave a few leaves behind the last full round of network workgroups,
    those leaves are put off to the next step (their slots keep the selected leaf).  Forced here with tiny rounds
    (dbaz_config.eval_round = round, eval_defer_max = largest left-over that is put off): the 96 games' rows, visit counts and statistics are
    bit-identical to the run without the cut; only the number of steps differs."""
    import torch
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd import nn as dnn
    torch.manual_seed(4)
    model = dnn.ResNetZero(dnn.resnet_params(3, 3, 32, 2, 4, 8))
    out, steps, exps = [], [], []
    for er, ed in ((-1, 0), rnd):
        e = Engine(3, 3, 40, mcts_num_read=48, noise=(0.8, 0.25), reuse_tree=True, evaluator="resnet", seed=11, eval_round=er,
                   eval_defer_max=ed)
        e.load_state_dict(model.state_dict(), "resnet", **model.shape)
        e.selfplay_start(96, 0)
        e.run()
        c = e.counters()
        assert c["games_finished"] == 96 and c["error_slots"] == 0 and c["active_slots"] == 0
        out.append(e.fetch_samples())
        steps.append(c["steps"])
        exps.append((c["expansions"], c["nn_evals"], c["cache_hits"], c["terminal_leaves"]))
        e.close()
    a, b = out
    assert len(a["z"]) == len(b["z"]) and exps[0] == exps[1]
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert steps[1] > steps[0]        # leaves were put off (and the games still came out the same)


def test_tight_node_pool_falls_back_to_a_fresh_root_bit_exactly():
    """The reference's trees are unbounded Python objects; a slot's node pool is not.  When the subtree kept by tree reuse
    (at most min(the chosen child's visits, the tree's nodes - 1) nodes) plus the next search (mcts_num_read + 2) would not fit nodes_per_slot, the
    driver starts that move from a fresh root -- exactly init_mcts_tree(..., reuse_tree=False) (mcts.py:176-179) -- and counts
    it.  The oracle replays every game move by move with that rule restated here: visit vectors, tree_size and the
    expansion count are bit-identical; with the default pool size the same games never need it."""
    import math
    from dotsboxesaz_amd.engine import Engine
    rows, cols, sims, cap, n_games = 3, 3, 40, 100, 60
    e = Engine(rows, cols, 16, mcts_num_read=sims, noise=(0.0, 0.0), evaluator="formula", seed=77, nodes_per_slot=cap)
    e.selfplay_start(n_games, 0)
    e.run()
    cnt = e.counters()
    assert cnt["games_finished"] == n_games and cnt["error_slots"] == 0 and cnt["pool_resets"] > 0
    assert cnt["pool_high_water"] <= cap
    got = e.fetch_samples()
    e.close()
    d = O.dims(rows, cols)
    ev = O.Evaluator(0)
    resets = searches = 0
    for gi in range(n_games):
        r = np.nonzero(got["game_idx"] == gi)[0]
        t = O.Tree(d, O.new_state(d))
        for i in r:
            n_valid = int(O.valid_moves(d, t.state).sum())
            reads = min(4 * math.factorial(n_valid), sims) if n_valid <= 5 else sims
            searches += reads + (0 if t.is_expanded else 1)
            t.search(reads, ev)
            _, _, nv, _ = t.root_arrays()
            assert np.array_equal(nv, got["visits"][i]), (gi, i)
            assert t.stats()[1] == got["tree_size"][i]
            mv = int(got["played"][i])
            # the rule of csrc/tree.hip reroot(): kept nodes <= min(visits of the child, nodes of the whole tree - 1)
            keep = min(int(nv[mv]), t.live_nodes() - 1) + sims + 2 <= cap
            resets += 0 if keep else 1
            t.advance(mv, reuse_tree=keep)
    assert resets == cnt["pool_resets"] and searches == cnt["expansions"]
    # the default pool (10 * (reads + 2) nodes) plays the same seeds without a single reset
    e = Engine(rows, cols, 16, mcts_num_read=sims, noise=(0.0, 0.0), evaluator="formula", seed=77)
    e.selfplay_start(n_games, 0)
    e.run()
    assert e.counters()["pool_resets"] == 0
    e.close()

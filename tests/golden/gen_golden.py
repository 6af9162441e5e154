#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Runs only in the build container (the reference lives at /root/reference and
never travels to the GPU box).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [--only rules,mcts,nn,selfplay]

Outputs are data only (inputs + the reference's outputs):
    rules.npz      seeded random playouts, BoxesState fields after every ply   (G1-G6)
    boards_csv.npz the 34 move sequences of the reference's test/test_boards.csv
                   with the resulting states
    mcts.npz       sequential UCT_search (max_pending_evals=1) root arrays/stats
                   under a formula-defined evaluator                           (M1-M9)
    pending.npz    UCT_search with max_pending_evals = K > 1 and an evaluator that
                   suspends once per call (synchronous waves)                   (8f-4)
    nn.npz         ResNetZero / SimpleNN outputs on committed / seeded weights  (N1-N3)
    selfplay.npz   SelfPlay.play_game + get_datasets with recorded RNG draws    (D1-D3)
    match.npz      two-model match play rows (self_play.compute_elo's loop)     (8f-3)
    train.npz      SymmetriesGenerator, dataset build, DataLoader epoch, train  (8f-1)
"""
import argparse
import copy
import asyncio
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DBAZ_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
warnings.simplefilter("ignore")

import torch  # noqa: E402
from dots_boxes.dots_boxes_game import BoxesState  # noqa: E402  (reference)
import mcts as ref_mcts  # noqa: E402  (reference)
from oracle import nn_ref  # noqa: E402  (only for randomize_bn / checksum helpers)

M64 = (1 << 64) - 1


def set_board(rows, cols):
    BoxesState.init_static_fields(((rows, cols),))


# ---------------------------------------------------------------- formula evaluator
def fmix64(x):
    x ^= x >> 33
    x = (x * 0xFF51AFD7ED558CCD) & M64
    x ^= x >> 33
    x = (x * 0xC4CEB9FE1A85EC53) & M64
    x ^= x >> 33
    return x


def played_bits(state):
    """Full-width edge bitmask (sum of 1<<move over played moves).  The reference keeps
    this in BoxesState.hash[0], but `1 << move` wraps at 64 bits whenever `move` is a
    numpy integer (np.argmax / np.random.choice results), so it is recomputed here from
    the board with python ints."""
    b = 0
    for i, e in enumerate(state.board.ravel().tolist()):
        if e == 255:
            b |= 1 << i
    return b


def formula_eval(state, kind):
    A = state.get_actions_size()
    if kind == 1:
        return np.ones(A, dtype=np.float32), np.array([0.0], dtype=np.float32)
    bits = played_bits(state)
    b2c2 = int(round(state.boxes_to_close[state.to_play] * 2))
    h = 0x9E3779B97F4A7C15
    for w in range(4):
        h = fmix64(h ^ ((bits >> (64 * w)) & M64))
    h = fmix64(h ^ ((b2c2 + 512) & M64))
    p = np.zeros(A, dtype=np.float32)
    for i in range(A):
        t = fmix64((h + (i + 1) * 0x9E3779B97F4A7C15) & M64)
        u = (t >> 20) & 0xFFFF
        p[i] = np.float32((u & 0xFF) + 1) * np.float32(((u >> 8) & 0xFF) + 1)
    t = fmix64(h ^ 0xD6E8FEB86659FD93)
    u = (t >> 17) & 0xFFFF
    v = (np.float32(u) - np.float32(32768.0)) / np.float32(32768.0)
    return p, np.array([v], dtype=np.float32)


def make_async_formula(kind):
    async def nn(state):
        return formula_eval(state, kind)
    return nn


def run(coro):
    loop = asyncio.new_event_loop()
    try:
        return loop.run_until_complete(coro)
    finally:
        loop.close()


def state_record(s):
    return dict(board=s.board.copy().ravel(), to_play=s.to_play,
                just_played=-1 if s.just_played is None else s.just_played,
                b2c2=[int(round(2 * s.boxes_to_close[0])), int(round(2 * s.boxes_to_close[1]))],
                result=2 if s.get_result() is None else s.get_result(),
                valid=s.get_valid_moves().copy(), features=s.get_features().ravel().copy(),
                hash_b2c2=int(round(2 * s.hash[1])),
                hash_words=[(played_bits(s) >> (64 * w)) & M64 for w in range(4)])


# ---------------------------------------------------------------- rules
def gen_rules():
    out = {}
    rng = np.random.RandomState(1234)
    boards = [(3, 3), (6, 6), (9, 9), (2, 3), (1, 1), (4, 2)]
    for (r, c) in boards:
        set_board(r, c)
        n_games = 6 if r * c > 1 else 1
        for gi in range(n_games):
            s = BoxesState()
            recs = [state_record(s)]
            moves, closed_n, closed_lc = [], [], []
            while True:
                vm = s.get_valid_moves(as_indices=True)
                if len(vm) == 0:
                    break
                # half of the games stop when the reference reports a result, the
                # rest play the board out completely (rules stay defined past the end)
                if gi % 2 == 0 and s.get_result() is not None:
                    break
                m = int(vm[rng.randint(len(vm))])
                cl = s.play_(m)
                moves.append(m)
                closed_n.append(len(cl))
                pad = [x for lc in cl for x in lc] + [-1] * (4 - 2 * len(cl))
                closed_lc.append(pad)
                recs.append(state_record(s))
            key = "b%dx%d_g%d" % (r, c, gi)
            out[key + "_moves"] = np.array(moves, dtype=np.int16)
            out[key + "_closed_n"] = np.array(closed_n, dtype=np.int8)
            out[key + "_closed_lc"] = np.array(closed_lc, dtype=np.int8).reshape(-1, 4)
            for f in recs[0]:
                dt = {"board": np.uint8, "valid": np.bool_, "features": np.int16,
                      "hash_words": np.uint64}.get(f, np.int16)
                out[key + "_" + f] = np.array([rc[f] for rc in recs], dtype=dt)
        # an illegal-move case: replaying a played edge / playing a sentinel raises ValueError
        s = BoxesState()
        s.play_(0)
        illegal = []
        for m in (0, c, 2 * (r + 1) * (c + 1) - 1):
            try:
                s.play_(m)
                illegal.append(0)
            except ValueError:
                illegal.append(1)
        out["b%dx%d_illegal" % (r, c)] = np.array(illegal, dtype=np.int8)
    out["boards"] = np.array(boards, dtype=np.int16)
    np.savez_compressed(os.path.join(HERE, "rules.npz"), **out)
    print("rules.npz", len(out), "arrays")


def load_csv():
    rows = []
    with open(os.path.join(REF, "test", "test_boards.csv")) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#") or line.startswith("id;"):
                continue
            parts = line.split(";")
            rows.append((int(parts[0]), [int(x) for x in parts[1].split()],
                         [int(x) for x in parts[2].split()], int(parts[3])))
    return rows


def gen_boards_csv():
    set_board(3, 3)
    rows = load_csv()
    out = {"ids": np.array([r[0] for r in rows], dtype=np.int16),
           "z": np.array([r[3] for r in rows], dtype=np.int8)}
    for (i, mv, nxt, _z) in rows:
        s = BoxesState()
        for m in mv:
            s.play_(m)
        rec = state_record(s)
        k = "id%d_" % i
        out[k + "moves"] = np.array(mv, dtype=np.int16)
        out[k + "next_moves"] = np.array(nxt, dtype=np.int16)
        out[k + "board"] = rec["board"]
        out[k + "features"] = rec["features"]
        out[k + "meta"] = np.array([rec["to_play"], rec["just_played"], rec["b2c2"][0], rec["b2c2"][1],
                                    rec["result"]], dtype=np.int16)
    np.savez_compressed(os.path.join(HERE, "boards_csv.npz"), **out)
    print("boards_csv.npz", len(rows), "positions")


# ---------------------------------------------------------------- mcts
def root_record(node):
    st = node.get_tree_stats()
    tv = node.total_value  # TreeRoot slot keyed by node.move (None only for a never re-rooted tree)
    return dict(priors=np.asarray(node.child_priors, dtype=np.float64).copy(),
                priors_is_f64=int(np.asarray(node.child_priors).dtype == np.float64),
                total_value=node.child_total_value.copy(), visits=node.child_number_visits.copy(),
                changed=node.child_player_changed.copy(),
                stats_i=np.array([st.max_deepness, st.tree_size, st.terminal_count], dtype=np.int32),
                q=np.float32(st.q_value),
                root_tv=np.float32(np.asarray(tv).ravel()[0]),
                root_nv=np.int32(node.number_visits))


def gen_mcts():
    out = {}
    cases = []

    def add_case(name, rows, cols, start_moves, kind, script, cpuct=(1.25, 19652)):
        """script: list of ('search', n, alpha, coeff) / ('advance', move or -1 for argmax, reuse)"""
        set_board(rows, cols)
        s = BoxesState()
        for m in start_moves:
            s.play_(m)
        node = ref_mcts.create_root_uct_node(s)
        nn = make_async_formula(kind)
        import zlib
        rng = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
        A = s.get_actions_size()
        steps = []
        si = 0
        for op in script:
            if op[0] == "search":
                _, n, alpha, coeff = op
                noise = None
                if alpha > 0:
                    noise = rng.dirichlet(np.full(A, alpha))
                    orig = np.random.dirichlet
                    np.random.dirichlet = lambda a, size=None, _n=noise: _n.reshape(1, -1).copy()
                try:
                    vc = run(ref_mcts.UCT_search(node, n, nn, cpuct, 1, (alpha, coeff)))
                finally:
                    if alpha > 0:
                        np.random.dirichlet = orig
                rec = root_record(node)
                assert np.array_equal(vc, rec["visits"])
                k = "%s_s%d_" % (name, si)
                for f, v in rec.items():
                    out[k + f] = np.asarray(v)
                out[k + "noise"] = noise if noise is not None else np.zeros(0)
                steps.append((0, n, alpha, coeff))
            else:
                _, mv, reuse = op
                if mv < 0:
                    mv = int(np.argmax(node.child_number_visits))
                node = ref_mcts.init_mcts_tree(node, mv, reuse_tree=bool(reuse))
                steps.append((1, mv, float(reuse), 0.0))
            si += 1
        out[name + "_script"] = np.array(steps, dtype=np.float64)
        out[name + "_cfg"] = np.array([rows, cols, kind, cpuct[0], cpuct[1]], dtype=np.float64)
        out[name + "_start"] = np.array(start_moves, dtype=np.int16)
        cases.append(name)

    S = lambda n, a=0.0, c=0.0: ("search", n, a, c)  # noqa: E731
    ADV = lambda mv=-1, reuse=1: ("advance", mv, reuse)  # noqa: E731
    for n in (1, 2, 25, 100, 800):
        add_case("e33_k0_n%d" % n, 3, 3, [], 0, [S(n)])
    add_case("e33_k1_n25", 3, 3, [], 1, [S(25)])
    add_case("e33_k1_n200", 3, 3, [], 1, [S(200)])
    csv = {r[0]: r for r in load_csv()}
    for cid in (1, 2, 3, 4, 5, 6, -1, -5, -10, -20, -24):
        add_case("csv%s_k0_n100" % str(cid).replace("-", "m"), 3, 3, csv[cid][1], 0, [S(100)])
    add_case("csv5_k1_n300", 3, 3, csv[5][1], 1, [S(300)])
    add_case("e66_k0_n25", 6, 6, [], 0, [S(25)])
    add_case("e66_k0_n800", 6, 6, [], 0, [S(800)])
    add_case("e66_k1_n200", 6, 6, [], 1, [S(200)])
    add_case("e99_k0_n200", 9, 9, [], 0, [S(200)])
    add_case("e23_k0_n100", 2, 3, [], 0, [S(100)])
    add_case("e11_k0_n30", 1, 1, [], 0, [S(30)])
    # tree reuse / fresh-root sequences, repeated searches on one root, injected noise
    add_case("seq33_reuse", 3, 3, [], 0, [S(100), ADV(), S(100), ADV(), S(60), ADV(), S(100), ADV(), S(100), ADV(),
                                          S(100), ADV(), S(100), ADV(), S(100)])
    add_case("seq33_fresh", 3, 3, [], 0, [S(80), ADV(-1, 0), S(80), ADV(-1, 0), S(80), ADV(-1, 0), S(80)])
    add_case("seq33_noise", 3, 3, [], 0, [S(50, 0.8, 0.25), ADV(), S(50, 0.8, 0.25), ADV(), S(50, 0.8, 0.25),
                                          ADV(), S(50, 0.8, 0.25)])
    add_case("seq33_repeat", 3, 3, [], 0, [S(40), S(40), S(40, 0.8, 0.25), S(40, 0.8, 0.25), S(40)])
    add_case("seq33_unvisited", 3, 3, [], 0, [S(10), ADV(27, 1), S(10), ADV(26, 0), S(10)])
    add_case("seq66_reuse", 6, 6, [], 0, [S(200), ADV(), S(200), ADV(), S(200), ADV(), S(200)])
    add_case("seq66_noise", 6, 6, [], 0, [S(150, 0.8, 0.25), ADV(), S(150, 0.8, 0.25), ADV(), S(150, 0.8, 0.25)])
    add_case("seq99_reuse", 9, 9, [], 0, [S(120, 0.8, 0.25), ADV(), S(120), ADV(), S(120)])
    add_case("cp33", 3, 3, [], 0, [S(120)], cpuct=(2.0, 500))
    # endgame: searches hit the 4*n! rule region and terminal leaves dominate
    add_case("end33", 3, 3, csv[4][1], 0, [S(60), ADV(), S(60), ADV(), S(24), ADV(), S(8)])
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "mcts.npz"), **out)
    print("mcts.npz", len(cases), "cases")


# ---------------------------------------------------------------- nn
def gen_pending():
    """UCT_search with max_pending_evals = K > 1 (mcts.py:228-239) under an evaluator that suspends exactly once
    (`await asyncio.sleep(0)`): the K searches of a wave then select one after the other, and expand + back up one after
    the other when the event loop comes round again -- the schedule the build's k_select_multi / k_expand_backup_multi
    implement (with virtual_visits = 0)."""
    out = {}
    cases = []

    def make_nn(kind):
        async def nn(state):
            await asyncio.sleep(0)
            return formula_eval(state, kind)
        return nn

    def add_case(name, rows, cols, start_moves, kind, K, script, cpuct=(1.25, 19652)):
        set_board(rows, cols)
        s = BoxesState()
        for m in start_moves:
            s.play_(m)
        node = ref_mcts.create_root_uct_node(s)
        nn = make_nn(kind)
        import zlib
        rng = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
        A = s.get_actions_size()
        steps = []
        si = 0
        for op in script:
            if op[0] == "search":
                _, n, alpha, coeff = op
                noise = None
                if alpha > 0:
                    noise = rng.dirichlet(np.full(A, alpha))
                    orig = np.random.dirichlet
                    np.random.dirichlet = lambda a, size=None, _n=noise: _n.reshape(1, -1).copy()
                try:
                    vc = run(ref_mcts.UCT_search(node, n, nn, cpuct, K, (alpha, coeff)))
                finally:
                    if alpha > 0:
                        np.random.dirichlet = orig
                rec = root_record(node)
                assert np.array_equal(vc, rec["visits"])
                k = "%s_s%d_" % (name, si)
                for f, v in rec.items():
                    out[k + f] = np.asarray(v)
                out[k + "noise"] = noise if noise is not None else np.zeros(0)
                steps.append((0, n, alpha, coeff))
            else:
                _, mv, reuse = op
                if mv < 0:
                    mv = int(np.argmax(node.child_number_visits))
                node = ref_mcts.init_mcts_tree(node, mv, reuse_tree=bool(reuse))
                steps.append((1, mv, float(reuse), 0.0))
            si += 1
        out[name + "_script"] = np.array(steps, dtype=np.float64)
        out[name + "_cfg"] = np.array([rows, cols, kind, cpuct[0], cpuct[1], K], dtype=np.float64)
        out[name + "_start"] = np.array(start_moves, dtype=np.int16)
        cases.append(name)

    S = lambda n, a=0.0, c=0.0: ("search", n, a, c)  # noqa: E731
    ADV = lambda mv=-1, reuse=1: ("advance", mv, reuse)  # noqa: E731
    csv = {r[0]: r for r in load_csv()}
    for K in (2, 4, 8, 64):
        add_case("p33_K%d_n100" % K, 3, 3, [], 0, K, [S(100)])
    add_case("p33_K8_n7", 3, 3, [], 0, 8, [S(7)])               # fewer reads than one wave
    add_case("p33_K64_n65", 3, 3, [], 0, 64, [S(65)])             # first wave min(K, A) = 32, then 33
    add_case("p33_K5_uniform", 3, 3, [], 1, 5, [S(120)])
    add_case("p33_K16_csv5", 3, 3, csv[5][1], 0, 16, [S(150)])
    add_case("p33_K8_end", 3, 3, csv[4][1], 0, 8, [S(60), ADV(), S(60), ADV(), S(24)])   # terminal leaves in the waves
    add_case("p33_K8_seq", 3, 3, [], 0, 8, [S(100), ADV(), S(100), ADV(), S(60, 0.8, 0.25), ADV(), S(100)])
    add_case("p33_K8_fresh", 3, 3, [], 0, 8, [S(80), ADV(-1, 0), S(80)])
    add_case("p66_K8_n200", 6, 6, [], 0, 8, [S(200)])
    add_case("p66_K64_seq", 6, 6, [], 0, 64, [S(300), ADV(), S(300, 0.8, 0.25), ADV(), S(200)])
    add_case("p99_K64_n400", 9, 9, [], 0, 64, [S(400)])
    add_case("p23_K16_n150", 2, 3, [], 0, 16, [S(150)])
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "pending.npz"), **out)
    print("pending.npz", len(cases), "cases")


def ref_resnet_params(rows, cols, channels, blocks, head_ch, value_fc, n_groups=1):
    from utils.utils import DotDict
    H, W = rows + 1, cols + 1
    return DotDict({"nn": {"model_parameters": {
        "resnet": {"pad_layer0": True, "in_channels": 3, "nb_channels": channels, "inner_channels": None,
                   "kernel_size": 3, "nb_blocks": blocks, "n_groups": n_groups},
        "policy_head": {"in_channels": channels, "inner_channels": head_ch, "fc_in": head_ch * H * W,
                        "nb_actions": 2 * H * W},
        "value_head": {"in_channels": channels, "inner_channels": head_ch, "fc_in": head_ch * H * W,
                       "fc_inner": value_fc}}, "pytorch_device": "cpu"}})


def sample_features(rows, cols, n, seed):
    """features of positions reached by seeded random playouts (real NN inputs)"""
    set_board(rows, cols)
    rng = np.random.RandomState(seed)
    X = []
    while len(X) < n:
        s = BoxesState()
        while s.get_result() is None and len(X) < n:
            if rng.rand() < 0.5:
                X.append(s.get_features().copy())
            vm = s.get_valid_moves(as_indices=True)
            s.play_(int(vm[rng.randint(len(vm))]))
    return np.stack(X, axis=0).astype(np.int16)


def gen_nn():
    import nn as ref_nn
    from dots_boxes.dots_boxes_nn import SimpleNN
    out = {}
    torch.set_num_threads(1)
    # (a) reduced ResNetZero with COMMITTED weights, three boards
    for (r, c, ch, nb, hc, vf, tag) in ((3, 3, 16, 2, 4, 8, "small33"), (6, 6, 16, 2, 4, 8, "small66"),
                                        (2, 3, 8, 1, 4, 4, "small23")):
        torch.manual_seed(7)
        params = ref_resnet_params(r, c, ch, nb, hc, vf)
        model = ref_nn.ResNetZero(params)
        nn_ref.randomize_bn(model, 11)
        wrapper = ref_nn.NeuralNetWrapper(model, params)
        X = sample_features(r, c, 12, 5)
        p, v = wrapper.predict_sync(X)
        out[tag + "_cfg"] = np.array([r, c, ch, nb, hc, vf], dtype=np.int32)
        out[tag + "_X"] = X
        out[tag + "_p"] = p
        out[tag + "_v"] = v
        for k, t in model.state_dict().items():
            out[tag + "_w_" + k] = t.numpy()
    # (a') the reference's n_groups option (nn.py:33-47,61-71): grouped 3x3 convs in the blocks, committed weights
    for (r, c, ch, nb, hc, vf, ng, tag) in ((3, 3, 16, 2, 4, 8, 2, "groups33"), (2, 3, 16, 1, 4, 4, 4, "groups23")):
        torch.manual_seed(13)
        params = ref_resnet_params(r, c, ch, nb, hc, vf, ng)
        model = ref_nn.ResNetZero(params)
        nn_ref.randomize_bn(model, 17)
        wrapper = ref_nn.NeuralNetWrapper(model, params)
        X = sample_features(r, c, 12, 6)
        p, v = wrapper.predict_sync(X)
        out[tag + "_cfg"] = np.array([r, c, ch, nb, hc, vf, ng], dtype=np.int32)
        out[tag + "_X"] = X
        out[tag + "_p"] = p
        out[tag + "_v"] = v
        for k, t in model.state_dict().items():
            out[tag + "_w_" + k] = t.numpy()
    # (b) full-size ResNetZero 20x64 -- weights regenerated from the seed, only I/O stored
    for (r, c, tag) in ((3, 3, "full33"), (6, 6, "full66"), (9, 9, "full99")):
        torch.manual_seed(0)
        params = ref_resnet_params(r, c, 64, 20, 16, 8)
        model = ref_nn.ResNetZero(params)
        nn_ref.randomize_bn(model, 3)
        wrapper = ref_nn.NeuralNetWrapper(model, params)
        X = sample_features(r, c, 8, 9)
        p, v = wrapper.predict_sync(X)
        out[tag + "_cfg"] = np.array([r, c, 64, 20, 16, 8], dtype=np.int32)
        out[tag + "_X"] = X
        out[tag + "_p"] = p
        out[tag + "_v"] = v
        out[tag + "_checksum"] = np.float64(nn_ref.state_dict_checksum(model))
    # (c) SimpleNN (3x3 only), seed-regenerated weights
    set_board(3, 3)
    torch.manual_seed(0)
    model = SimpleNN(None)
    nn_ref.randomize_bn(model, 3)
    from utils.utils import DotDict
    wrapper = ref_nn.NeuralNetWrapper(model, DotDict({"nn": {"pytorch_device": "cpu"}}))
    X = sample_features(3, 3, 8, 9)
    p, v = wrapper.predict_sync(X)
    out["simple_X"] = X
    out["simple_p"] = p
    out["simple_v"] = v
    out["simple_checksum"] = np.float64(nn_ref.state_dict_checksum(model))
    np.savez_compressed(os.path.join(HERE, "nn.npz"), **out)
    print("nn.npz", len(out), "arrays")


# ---------------------------------------------------------------- self-play
def gen_selfplay():
    import self_play as ref_sp
    import nn as ref_nn
    from utils.utils import DotDict
    out = {}
    cases = []

    def run_case(name, rows, cols, sims, noise, reuse, n_games, seed, evaluator, temperature=None):
        set_board(rows, cols)
        temperature = {0: 1.0, 12: 0.02} if temperature is None else temperature
        params = DotDict({"self_play": {"reuse_mcts_tree": bool(reuse), "noise": list(noise),
                                        "mcts": {"mcts_num_read": sims, "mcts_cpuct": [1.25, 19652],
                                                 "temperature": dict(temperature), "max_async_searches": 1}}})
        np.random.seed(seed)
        drawn_noise, drawn_moves = [], []
        o_dir, o_ch = np.random.dirichlet, np.random.choice

        def rec_dir(alpha, size=None):
            r = o_dir(alpha, size)
            drawn_noise.append(np.asarray(r).ravel().copy())
            return r

        def rec_choice(a, size=None, replace=True, p=None):
            r = o_ch(a, size, replace, p)
            drawn_moves.append(int(np.asarray(r).ravel()[0]))
            return r

        np.random.dirichlet, np.random.choice = rec_dir, rec_choice
        try:
            sp = ref_sp.SelfPlay(evaluator, params)
            run(sp.play_games(BoxesState(), list(range(n_games))))
        finally:
            np.random.dirichlet, np.random.choice = o_dir, o_ch
        df = sp.get_datasets(3, with_features=True).reset_index()
        A = 2 * (rows + 1) * (cols + 1)
        F = 3 * (rows + 1) * (cols + 1)
        k = name + "_"
        out[k + "cfg"] = np.array([rows, cols, sims, noise[0], noise[1], int(reuse), n_games, seed], dtype=np.float64)
        out[k + "temp"] = np.array(sorted(temperature.items()), dtype=np.float64)
        out[k + "index"] = df[["generation", "game_idx", "move_idx"]].to_numpy().astype(np.int16)
        out[k + "move"] = df["move"].to_numpy().astype(np.int16)
        out[k + "player"] = df["player"].to_numpy().astype(np.int8)
        out[k + "x"] = df[["x_%d" % i for i in range(F)]].to_numpy().astype(np.int16)
        out[k + "pi"] = df[["pi_%d" % i for i in range(A)]].to_numpy().astype(np.float64)
        out[k + "z"] = df["z"].to_numpy().astype(np.int64)
        out[k + "stats"] = df[["max_deepness", "tree_size", "terminal_count"]].to_numpy().astype(np.int32)
        out[k + "q"] = df["q_value"].to_numpy().astype(np.float32)
        out[k + "drawn_moves"] = np.array(drawn_moves, dtype=np.int16)
        out[k + "drawn_noise"] = (np.stack(drawn_noise) if drawn_noise else np.zeros((0, A)))
        out[k + "dtypes"] = np.array([str(df[c].dtype) for c in ("move", "player", "x_0", "pi_0", "z", "max_deepness",
                                                                    "tree_size", "terminal_count", "q_value")])
        cases.append(name)
        print("  ", name, "rows", len(df))

    run_case("sp33_formula_noise", 3, 3, 25, (0.8, 0.25), True, 2, 0, make_async_formula(0))
    run_case("sp33_formula_fresh", 3, 3, 25, (0.0, 0.0), False, 2, 1, make_async_formula(0))
    run_case("sp33_uniform", 3, 3, 40, (0.8, 0.25), True, 1, 2, make_async_formula(1))
    run_case("sp66_formula", 6, 6, 100, (0.8, 0.25), True, 1, 3, make_async_formula(0))
    run_case("sp23_formula", 2, 3, 60, (0.8, 0.25), True, 2, 4, make_async_formula(0), temperature={0: 1.0, 4: 0.02})

    # BASELINE config 1: 3x3, 1 game, 25 sims, random-init ResNetZero (reduced size so that
    # the per-leaf outputs can be committed); every (hash -> p, v) the reference's net
    # returned is recorded so that the test evaluator is bit-exact on any host.
    set_board(3, 3)
    torch.set_num_threads(1)
    torch.manual_seed(7)
    params = ref_resnet_params(3, 3, 16, 2, 4, 8)
    model = ref_nn.ResNetZero(params)
    nn_ref.randomize_bn(model, 11)
    wrapper = ref_nn.NeuralNetWrapper(model, params)
    log_k, log_p, log_v = [], [], []

    async def net(state):
        p, v = wrapper.predict_sync(np.stack([state.get_features()], axis=0))
        bits = played_bits(state)
        log_k.append([(bits >> (64 * w)) & M64 for w in range(4)] +
                     [(int(round(2 * state.boxes_to_close[state.to_play])) + 512) & M64])
        log_p.append(p[0].copy())
        log_v.append(v[0].copy())
        return p[0], v[0]

    run_case("sp33_resnet", 3, 3, 25, (0.8, 0.25), True, 1, 0, net)
    out["sp33_resnet_evalkeys"] = np.array(log_k, dtype=np.uint64)
    out["sp33_resnet_evalp"] = np.stack(log_p).astype(np.float32)
    out["sp33_resnet_evalv"] = np.stack(log_v).astype(np.float32)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "selfplay.npz"), **out)
    print("selfplay.npz", len(cases), "cases")


# ---------------------------------------------------------------- match play / Elo
def gen_match():
    """self_play.compute_elo's game loop: SelfPlay with a player_change_callback that swaps the
    evaluator by root.to_play (self_play.py:59,237-239), no tree reuse, no noise, rows without
    features (get_datasets(generations, with_features=False)), and utils.elo_rating2."""
    import self_play as ref_sp
    from utils.utils import DotDict, elo_rating2
    out = {}
    cases = []
    for (name, rows, cols, sims, n_games, seed) in (("m33", 3, 3, 30, 4, 5), ("m23", 2, 3, 40, 3, 6)):
        set_board(rows, cols)
        params = DotDict({"self_play": {"reuse_mcts_tree": False, "noise": [0.0, 0.0],
                                        "mcts": {"mcts_num_read": sims, "mcts_cpuct": [1.25, 19652],
                                                 "temperature": {0: 1.0, 12: 0.02}, "max_async_searches": 1}}})
        np.random.seed(seed)
        state = {"model": 0, "game": 0}

        async def nn(st, _s=state):
            return formula_eval(st, 0 if _s["model"] == 0 else 1)  # model 0: hash formula, model 1: uniform

        drawn = []
        o_ch = np.random.choice

        def rec_choice(a, size=None, replace=True, p=None):
            r = o_ch(a, size, replace, p)
            drawn.append(int(np.asarray(r).ravel()[0]))
            return r

        np.random.choice = rec_choice
        try:
            sp = ref_sp.SelfPlay(nn, params)
            # seats are swapped on odd games (the reference swaps them by worker pid, self_play.py:202-209)
            sp.set_player_change_callback(lambda player, _s=state: _s.__setitem__("model", player ^ (_s["game"] & 1)))
            for gi in range(n_games):
                state["game"] = gi
                run(sp.play_game(BoxesState(), gi))
        finally:
            np.random.choice = o_ch
        df = sp.get_datasets([7, 9], with_features=False).reset_index()
        assert not any(c.startswith("x_") for c in df.columns)
        A = 2 * (rows + 1) * (cols + 1)
        k = name + "_"
        out[k + "cfg"] = np.array([rows, cols, sims, n_games, seed], dtype=np.int32)
        out[k + "index"] = df[["generation", "game_idx", "move_idx"]].to_numpy().astype(np.int16)
        out[k + "move"] = df["move"].to_numpy().astype(np.int16)
        out[k + "player"] = df["player"].to_numpy().astype(np.int8)
        out[k + "pi"] = df[["pi_%d" % i for i in range(A)]].to_numpy().astype(np.float64)
        out[k + "z"] = df["z"].to_numpy().astype(np.int64)
        out[k + "stats"] = df[["max_deepness", "tree_size", "terminal_count"]].to_numpy().astype(np.int32)
        out[k + "q"] = df["q_value"].to_numpy().astype(np.float32)
        out[k + "drawn_moves"] = np.array(drawn, dtype=np.int16)
        out[k + "columns"] = np.array(list(df.columns))
        cases.append(name)
        print("  ", name, "rows", len(df))
    tuples = [(1000.0, 1000.0, 12, 8), (1200.0, 950.5, 3, 17), (0.0, 0.0, 0, 20), (1500.0, 1500.0, 10, 10), (-30.5, 400.0, 7, 0)]
    out["elo_in"] = np.array(tuples, dtype=np.float64)
    out["elo_out"] = np.array([elo_rating2(a, b, int(n0), int(n1), K=30) for (a, b, n0, n1) in tuples], dtype=np.float64)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "match.npz"), **out)
    print("match.npz", len(cases), "cases")


# ---------------------------------------------------------------- training data path
def gen_train():
    """SymmetriesGenerator (dots_boxes_nn.py:11-58) for each of its 8 transforms; the array build of
    HDFStoreDataset (utils/utils.py:66-80: training filter, df.sample, pos_average groupby-mean)
    applied to rows the reference's SelfPlay produced (the HDF read itself needs pytables, which the
    image lacks, so those pandas statements are executed here on the in-memory DataFrame); one
    shuffled DataLoader epoch with symmetries (nn.py:186-216); AlphaZeroLoss and
    NeuralNetWrapper.train on a small ResNetZero (nn.py:131-138,175-295)."""
    import random
    import tempfile
    import pandas as pd
    import self_play as ref_sp
    import nn as ref_nn
    from dots_boxes.dots_boxes_nn import SymmetriesGenerator
    from utils import utils as ref_utils
    from utils.utils import DotDict
    from torch.utils import data
    out = {}
    symm = SymmetriesGenerator()
    o_randint = random.randint
    # ---- (a) the 8 transforms
    for (rows, cols) in ((3, 3), (6, 6)):
        H, W = rows + 1, cols + 1
        rs = np.random.RandomState(100 + rows)
        boards = rs.randint(-3, 9, size=(5, 3, H, W)).astype(np.float32)
        pol = rs.rand(5, 2 * H * W).astype(np.float32)
        k = "sym%d%d_" % (rows, cols)
        out[k + "boards"], out[k + "pol"] = boards, pol
        for sym in range(8):
            random.randint = lambda a, b, _s=sym: _s
            try:
                b2, p2 = symm(torch.tensor(boards), torch.tensor(pol))
            finally:
                random.randint = o_randint
            out[k + "boards_out%d" % sym] = b2.numpy().copy()
            out[k + "pol_out%d" % sym] = p2.numpy().copy()
    # ---- (b) dataset build on reference self-play rows (3x3, 10 games, 25 sims)
    set_board(3, 3)
    params = DotDict({"self_play": {"reuse_mcts_tree": True, "noise": [0.8, 0.25],
                                    "mcts": {"mcts_num_read": 25, "mcts_cpuct": [1.25, 19652],
                                             "temperature": {0: 1.0, 6: 0.02}, "max_async_searches": 1}}})
    np.random.seed(21)
    sp = ref_sp.SelfPlay(make_async_formula(0), params)
    run(sp.play_games(BoxesState(), list(range(10))))
    df0 = sp.get_datasets(2, with_features=True)
    A, F = 32, 48
    xcols = ["x_%d" % i for i in range(F)]
    picols = ["pi_%d" % i for i in range(A)]
    # visit counts are not a column of the reference's table: recover them from the nodes the way
    # get_datasets does (self_play.py:113-115) so that the packed replay rows can be rebuilt
    vis = []
    for idx, nodes, z in sp.played_games:
        for n in nodes[:-1]:
            vis.append(np.asarray(n.child_number_visits, dtype=np.int64).copy())
    vis = np.stack(vis)
    assert np.array_equal(vis / vis.sum(1, keepdims=True), df0[picols].to_numpy())
    out["ds_x"] = df0[xcols].to_numpy().astype(np.int16)
    out["ds_visits"] = vis.astype(np.int32)
    out["ds_z"] = df0["z"].to_numpy().astype(np.int8)
    out["ds_game_idx"] = df0.reset_index()["game_idx"].to_numpy().astype(np.int32)
    out["ds_move_idx"] = df0.reset_index()["move_idx"].to_numpy().astype(np.int16)
    df0 = df0.assign(rowid=np.arange(len(df0)))
    # coach.py:59-65 (DataFrame.append of old pandas == concat)
    np.random.seed(22)
    train = df0.sample(frac=0.9).assign(training=1)
    val = df0[~df0.index.isin(train.index)].assign(training=-1)
    new_samples = pd.concat([train, val]).astype({"training": np.int8})
    for flag, name in ((1, "train"), (-1, "val")):
        for avg in (False, True):
            np.random.seed(23)
            df = new_samples[new_samples.training == flag]          # utils.py:67
            df = df.sample(min(int(1e12), df.shape[0]))             # utils.py:68
            order = df["rowid"].to_numpy().astype(np.int32)
            cols_ = df.columns
            features_cols = list(c for c in cols_ if c.startswith("x_"))
            if avg:
                df = df.groupby(features_cols).mean().reset_index()  # utils.py:72-73
            k = "ds_%s_%s_" % (name, "avg" if avg else "raw")
            out[k + "order"] = order
            out[k + "features"] = df[features_cols].values.astype(np.float32).reshape(-1, 3, 4, 4)
            out[k + "policy"] = df[list(c for c in cols_ if c.startswith("pi_"))].values.astype(np.float32)
            out[k + "value"] = df.z.values.astype(np.float32)
            print("  ", k, len(order), "->", len(df))
    # ---- (c) one DataLoader epoch: shuffle, drop_last, symmetries (nn.py:186-216)
    ds = ref_utils.HDFStoreDataset.__new__(ref_utils.HDFStoreDataset)
    data.Dataset.__init__(ds)
    ds.features = out["ds_train_avg_features"]
    ds.policy = out["ds_train_avg_policy"]
    ds.value = out["ds_train_avg_value"]
    torch.manual_seed(31)
    random.seed(32)
    drawn = []

    def rec_randint(a, b):
        r = o_randint(a, b)
        drawn.append(r)
        return r

    random.randint = rec_randint
    try:
        bs = 16
        loader = data.DataLoader(ds, bs, shuffle=True, drop_last=True)
        ep_b, ep_p, ep_z = [], [], []
        for epoch in range(2):
            for boards, pi, z in loader:
                boards, pi = symm(boards, pi)
                ep_b.append(boards.numpy().copy()); ep_p.append(pi.numpy().copy()); ep_z.append(z.numpy().copy())
    finally:
        random.randint = o_randint
    out["ld_boards"], out["ld_pi"], out["ld_z"] = np.stack(ep_b), np.stack(ep_p), np.stack(ep_z)
    out["ld_syms"] = np.array(drawn, dtype=np.int32)
    out["ld_cfg"] = np.array([31, 32, bs, 2], dtype=np.int64)
    # ---- (d) AlphaZeroLoss + NeuralNetWrapper.train (generation 1: min(2*1, nb_epochs) epochs)
    torch.set_num_threads(1)
    tmp = tempfile.mkdtemp(prefix="dbaz_golden_")
    p = ref_resnet_params(3, 3, 16, 2, 4, 8)
    p.nn.chkpts_filename = os.path.join(tmp, "model_gen{}.pt")
    p.nn.train_params = DotDict({"nb_epochs": 2, "train_batch_size": 16, "val_batch_size": 16, "lr": 1e-2,
                                 "optimizer_params": {"momentum": 0.9, "weight_decay": 1e-4},
                                 "symmetries": symm})
    torch.manual_seed(41)
    model = ref_nn.ResNetZero(p)
    nn_ref.randomize_bn(model, 42)
    opt0 = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    ref_nn.save_checkpoint(p.nn.chkpts_filename.format(0), model, opt0, 7)
    sd0 = {k_: v.detach().clone() for k_, v in model.state_dict().items()}
    for k_, v in sd0.items():
        out["tr_sd0/" + k_] = v.numpy()
    crit = ref_nn.AlphaZeroLoss()
    model.train(False)
    xb = torch.tensor(out["ld_boards"][0]); pib = torch.tensor(out["ld_pi"][0]); zb = torch.tensor(out["ld_z"][0])
    pp, vv = model(xb)
    loss, (lpi, lv) = crit(pp, vv, pib, zb)
    out["tr_loss_eval"] = np.array([loss.item(), lpi, lv], dtype=np.float64)

    class W:
        def __init__(self):
            self.scalars = []

        def add_scalar(self, tag, v, i):
            self.scalars.append((tag, float(v), int(i)))

        def add_scalars(self, tag, d, i):
            for k_, v in sorted(d.items()):
                self.scalars.append((tag + "/" + k_, float(v), int(i)))

    val_ds = ref_utils.HDFStoreDataset.__new__(ref_utils.HDFStoreDataset)
    data.Dataset.__init__(val_ds)
    # validation uses batches of 16 with drop_last: give it >= 16 rows (train rows reused)
    val_ds.features, val_ds.policy, val_ds.value = ds.features[:40], ds.policy[:40], ds.value[:40]
    wr = W()
    wrapper = ref_nn.NeuralNetWrapper(ref_nn.ResNetZero(p), p)
    torch.manual_seed(51)
    random.seed(52)
    last = wrapper.train(ds, val_ds, wr, 1)
    out["tr_last_batch_idx"] = np.array([last], dtype=np.int64)
    out["tr_scalar_tags"] = np.array([t for t, _, _ in wr.scalars])
    out["tr_scalar_vals"] = np.array([v for _, v, _ in wr.scalars], dtype=np.float64)
    out["tr_scalar_steps"] = np.array([i for _, _, i in wr.scalars], dtype=np.int64)
    ck = torch.load(p.nn.chkpts_filename.format(1), map_location="cpu", weights_only=True)
    for k_, v in ck["model_dict"].items():
        out["tr_sd1/" + k_] = v.numpy()
    out["tr_ck_keys"] = np.array(sorted(ck.keys()))
    out["tr_cfg"] = np.array([51, 52, 16, 2], dtype=np.int64)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, "train.npz"), **out)
    print("train.npz", len(out), "arrays")


# ---------------------------------------------------------------- training-mode residual blocks
def gen_train_tower():
    """The reference's own ResBlock modules (nn.py:33-58) under .train(True): forward on a post-ReLU input, backward of a
    random output gradient -- what csrc/train.hip (dbaz_trainer_*) computes.  Weights are regenerated from the seed in the test
    (same construction order as oracle/nn_ref._Block; checksum stored), BatchNorm affines / running statistics randomised.
    Seeds are advanced until no ReLU input of the float64 evaluation lies within 2e-6 of zero (a float32 evaluation may put
    such an element on the other side and the gradients are then not comparable)."""
    import nn as ref_nn
    out = {}
    torch.set_num_threads(1)
    for (r, c, nb, n, tag) in ((3, 3, 1, 6, "t33"), (6, 6, 2, 5, "t66")):
        H, W = r + 1, c + 1
        for seed in range(100, 200):
            torch.manual_seed(seed)
            blocks = torch.nn.Sequential(*[ref_nn.ResBlock(64, 3, 1, None) for _ in range(nb)])
            nn_ref.randomize_bn(blocks, seed + 1)
            g = torch.Generator().manual_seed(seed + 2)
            x = torch.relu(torch.randn(n, 64, H, W, generator=g))
            gout = torch.randn(n, 64, H, W, generator=g) * 1e-2
            b64 = copy.deepcopy(blocks).double().train(True)
            margin, xx = float("inf"), x.double()
            with torch.no_grad():
                for blk in b64:
                    pre1 = blk.bn1(blk.conv1(xx))
                    pre2 = blk.bn2(blk.conv2(torch.relu(pre1))) + xx
                    margin = min(margin, float(pre1.abs().min()), float(pre2.abs().min()))
                    xx = torch.relu(pre2)
            if margin >= 2e-6:
                break
        checksum = nn_ref.state_dict_checksum(blocks)
        blocks.train(True)
        xr = x.clone().requires_grad_(True)
        y = blocks(xr)
        y.backward(gout)
        out[tag + "_cfg"] = np.array([r, c, nb, n, seed], dtype=np.int32)
        out[tag + "_checksum"] = np.float64(checksum)
        out[tag + "_out"] = y.detach().numpy()
        out[tag + "_grad_x"] = xr.grad.numpy()
        for k, p_ in blocks.named_parameters():
            gnp = p_.grad.numpy()
            out[tag + "_g_" + k] = gnp if gnp.size <= 64 else gnp.ravel()[::37].copy()   # conv gradients: every 37th element
        for k, v in blocks.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[tag + "_s_" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "train_tower.npz"), **out)
    print("train_tower.npz", len(out), "arrays")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="rules,boards,mcts,pending,nn,selfplay,match,train,train_tower")
    args = ap.parse_args()
    todo = args.only.split(",")
    if "rules" in todo:
        gen_rules()
    if "boards" in todo:
        gen_boards_csv()
    if "mcts" in todo:
        gen_mcts()
    if "pending" in todo:
        gen_pending()
    if "nn" in todo:
        gen_nn()
    if "selfplay" in todo:
        gen_selfplay()
    if "match" in todo:
        gen_match()
    if "train" in todo:
        gen_train()
    if "train_tower" in todo:
        gen_train_tower()

#!/usr/bin/env python3
"""BASELINE configs[2] with the flagship network, checked bit for bit: 8 192 concurrent 6x6 games at 800 reads per move on
ResNetZero 20x64 (random init, f16x3), complete games with device-sampled moves and the noise off (numpy's Dirichlet stream
cannot be matched on the device); then K of the games are replayed by the oracle's sequential search (oracle/dbaz_oracle.c),
teacher-forced with the device's moves, its evaluator asking the same engine for (p, v) of one position at a time.  Every
row of those games -- visit counts, pi, q, TreeStats, z -- must be identical.  Test infrastructure (uses oracle/; not collected by pytest: ~3 min of GPU time):

    python tests/verify_headline_games.py [K] [slots] [sims]      -> one JSON line"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(k=2, slots=8192, sims=800):
    import numpy as np
    import torch
    from oracle import oracle as O
    from oracle import nn_ref
    from dotsboxesaz_amd.engine import Engine
    torch.manual_seed(0)
    m = nn_ref.ResNetZeroRef(6, 6, 64, 20)
    nn_ref.randomize_bn(m, 3)
    e = Engine(6, 6, slots, mcts_num_read=sims, noise=(0.0, 0.0), evaluator="resnet", seed=2, nn_precision=1)
    e.load_state_dict(m.state_dict(), "resnet", 64, 20, 16, 8)
    t0 = time.perf_counter()
    e.selfplay_start(slots, 0)
    e.run()
    t_play = time.perf_counter() - t0
    c = e.counters()
    assert c["games_finished"] == slots and c["error_slots"] == 0
    got = e.fetch_samples()
    d = O.dims(6, 6)
    memo = {}

    def hip_net(dd, st):
        x = O.features(dd, st)
        key = x.tobytes()
        if key not in memo:
            pv = e.predict(x.astype(np.float32).reshape(1, 3, 7, 7))
            memo[key] = (pv[0][0].copy(), pv[1][0].copy())
        return memo[key]

    ev = O.Evaluator(hip_net)
    pp = O.selfplay_params(sims, noise=(0.0, 0.0), reuse_tree=True)
    rows = 0
    t0 = time.perf_counter()
    games = [int(g) for g in np.linspace(0, slots - 1, k).astype(int)]
    for gi in games:
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
        rows += len(r)
    t_replay = time.perf_counter() - t0
    print(json.dumps({"what": "configs[2] with ResNetZero 20x64 (f16x3), noise off: complete games vs the oracle's sequential search "
                              "fed by the same network, bit-exact rows", "slots": slots, "sims": sims, "games_played": slots,
                      "games_per_sec": slots / t_play, "expansions": c["expansions"], "cache_hits": c["cache_hits"],
                      "pool_resets": c["pool_resets"], "games_replayed": games, "rows_compared": rows,
                      "oracle_network_calls": len(memo), "seconds_play": t_play, "seconds_replay": t_replay, "identical": True}))
    e.close()


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    main(*a)

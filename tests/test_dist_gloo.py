"""Multi-rank path on CPU (gloo, world_size 2): game sharding and the replay all-gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dotsboxesaz_amd.self_play import all_gather_rows, shard_games


def test_shard_games_is_array_split():
    for n, w in [(2000, 8), (7, 3), (5, 8), (8192, 8), (1, 2)]:
        ref = np.array_split(np.arange(n), w)
        for r in range(w):
            first, cnt = shard_games(n, w, r)
            assert cnt == len(ref[r]) and (cnt == 0 or first == ref[r][0])
        assert sum(shard_games(n, w, r)[1] for r in range(w)) == n


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rb = 40
    n = 5 + 3 * rank  # ragged shards
    rows = torch.full((n, rb), rank + 1, dtype=torch.uint8)
    rows[:, 0] = torch.arange(n, dtype=torch.uint8)
    allrows, counts = all_gather_rows(rows, dist)
    empty, c2 = all_gather_rows(torch.zeros((0, rb), dtype=torch.uint8), dist)
    ok = (counts == [5, 8] and allrows.shape == (13, rb) and bool((allrows[:5, 1] == 1).all())
          and bool((allrows[5:, 1] == 2).all()) and allrows[5:, 0].tolist() == list(range(8))
          and empty.shape[0] == 0 and c2 == [0, 0])
    out[rank] = int(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_replay_all_gather_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0, 0])
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert list(out) == [1, 1]

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


# ---- the driver contract under torch.distributed: every rank's games reach the DataFrame -------------
class _StubNN:
    kind = "formula"
    shape = {}

    def __init__(self, params):
        self.loaded = None

    def load_parameters(self, generation, to_device=None):
        self.loaded = generation  # generation g plays with g-1's weights (self_play.py:187-190)


class _FakeEngine:
    """Stands in for the HIP engine on CPU: geometry only (the rows come from _fake_collect)."""

    def __init__(self, rows, cols, n_slots, **kw):
        self.rows, self.cols = rows, cols
        self.F, self.A = 3 * (rows + 1) * (cols + 1), 2 * (rows + 1) * (cols + 1)

    def close(self):
        pass


def _fake_rows(first, count, F, A):
    """Packed replay rows (csrc/common.h RowMeta | x | visits) of games first..first+count-1,
    3 + game % 3 rows each, contents a function of (game, ply) only."""
    from dotsboxesaz_amd import self_play as sp
    rb = sp.row_bytes(F, A)
    rows = []
    for g in range(first, first + count):
        for ply in range(3 + g % 3):
            meta = np.zeros(1, sp.ROW_META)
            meta["game_idx"], meta["move_idx"], meta["move"], meta["played"] = g, ply, ply - 1, ply
            meta["player"], meta["z"], meta["q_value"], meta["tree_size"] = ply & 1, 1 - 2 * (ply & 1), g + ply / 8.0, 7 * g
            x = ((np.arange(F) + g + ply) % 2).astype("<i2")
            vis = ((np.arange(A) * 7 + g * 3 + ply) % 5).astype("<i4")
            raw = meta.tobytes() + x.tobytes() + vis.tobytes()
            rows.append(np.frombuffer(raw + b"\0" * (rb - len(raw)), np.uint8))
    return torch.from_numpy(np.stack(rows).copy()) if rows else torch.zeros((0, rb), dtype=torch.uint8)


def _gen_worker(rank, world, port, out, n_games):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dotsboxesaz_amd import engine as E
    from dotsboxesaz_amd import self_play as sp
    E.Engine = _FakeEngine
    sp.collect_rows_device = lambda eng, count, first: _fake_rows(first, count, eng.F, eng.A)
    df = sp.generate_games(None, 3, _StubNN, n_games, {}, rows=2, cols=2, dist=dist)
    ref = sp.samples_to_dataframe(sp.unpack_rows(_fake_rows(0, n_games, 27, 18).numpy(), 27, 18), 3, 2, 2, True)
    ref["training"] = np.zeros(len(ref.index), dtype=np.int8)
    games = df.index.get_level_values("game_idx")
    ok = (sorted(set(games)) == list(range(n_games)) and len(df) == sum(3 + g % 3 for g in range(n_games))
          and not df.index.duplicated().any() and df.equals(ref))
    out[rank] = int(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_generate_games_world2_returns_every_ranks_rows():
    """self_play.py:291-306 + :264-265: all workers' rows end up in the dataset.  Two gloo ranks,
    7 games (a ragged 4 + 3 split); both ranks must hold games 0..6 exactly once, equal to the
    single-process frame."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Array("i", [0, 0])
    procs = [ctx.Process(target=_gen_worker, args=(r, 2, port, out, 7)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert list(out) == [1, 1]


def test_unpack_rows_layout():
    from dotsboxesaz_amd import self_play as sp
    F, A = 48, 32
    s = sp.unpack_rows(_fake_rows(2, 3, F, A).numpy()[::-1], F, A)  # any order in, sorted out
    assert s["game_idx"].tolist() == [2] * 5 + [3] * 3 + [4] * 4
    assert s["move_idx"].tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 0, 1, 2, 3]
    assert s["pi"].dtype == np.float64 and np.allclose(s["pi"].sum(1), 1.0)
    assert np.array_equal(s["visits"][0], (np.arange(A) * 7 + 6) % 5) and np.array_equal(s["x"][0], (np.arange(F) + 2) % 2)
    assert s["q_value"][1] == np.float32(2.125) and s["z"].tolist()[:2] == [1, -1]
    empty = sp.unpack_rows(np.zeros((0, sp.row_bytes(F, A)), np.uint8), F, A)
    assert len(empty["z"]) == 0 and empty["pi"].shape == (0, A)

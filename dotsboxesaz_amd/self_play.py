"""Host-side mirror of the reference's self-play driver contract (self_play.py).

    generate_games(hdf_file_name, generation, nn_class, n_games, params, ...)   self_play.py:291-306
    SelfPlay(nn, params).play_games(...) / get_datasets(generation)             self_play.py:19-156

The games themselves run on the GPU (dotsboxesaz_amd.engine.Engine): thousands of concurrent
games, one sequential search per game (the reference's max_async_searches=1 semantics), slots
refilled as games finish.  This module only shapes configuration in and DataFrames out, shards
game indices over ranks and all-gathers the replay rows over RCCL at iteration end.
"""
import time

import numpy as np


def _get(d, k, default=None):
    if d is None:
        return default
    if isinstance(d, dict):
        return d.get(k, default)
    return getattr(d, k, default)


def shard_games(n_games, world_size, rank):
    """Contiguous game-index range of `rank` (np.array_split semantics, self_play.py:294)."""
    base, extra = divmod(int(n_games), int(world_size))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def engine_kwargs_from_params(params, async_searches=False):
    """Pull the hot-path knobs out of a reference-style params dict (configuration.py:82-100).

    async_searches: the reference runs every self-play search with `max_async_searches` simulations of the ONE tree in flight
    (self_play.py:27-30, configuration.py:99: 64), only to fill a GPU batch from a single game -- 98 % of those leaves are
    duplicates.  Here the batch comes from thousands of concurrent games, so the default is the sequential search per
    game (max_pending_evals = 1: the semantics of every parity path).  async_searches=True honours the knob instead: every search
    of every game runs in waves of self_play.mcts.max_async_searches simulations with the reference's bookkeeping
    (dbaz_config.selfplay_pending; equal to the reference's own UCT_search(..., max_pending_evals=K) under an evaluator that
    suspends once per call, tests/test_hip_pending.py)."""
    sp = _get(params, "self_play")
    m = _get(sp, "mcts")
    noise = _get(sp, "noise", [0.0, 0.0])
    kw = dict(mcts_num_read=int(_get(m, "mcts_num_read", 800)), cpuct=tuple(_get(m, "mcts_cpuct", (1.25, 19652))),
              noise=(float(noise[0]), float(noise[1])), temperature=dict(_get(m, "temperature", {0: 1.0})),
              reuse_tree=bool(_get(sp, "reuse_mcts_tree", True)))
    k = int(_get(m, "max_async_searches", 1) or 1)
    if async_searches and k > 1:
        kw.update(max_pending_evals=k, selfplay_pending=True, transposition_cache=False)
    return kw


def samples_to_dataframe(s, generation, rows, cols, with_features=True):
    """SelfPlay.get_datasets schema (self_play.py:95-156): index (generation, game_idx, move_idx);
    columns move i16, player i8, x_* i16, pi_* f64, z i64, max_deepness i16, tree_size i32,
    terminal_count i32, q_value f32."""
    import pandas as pd
    n = len(s["move"])
    A = s["pi"].shape[1]
    F = s["x"].shape[1]
    if isinstance(generation, (list, tuple)):
        gen = np.where(s["player"] == 0, generation[0], generation[1]).astype(np.int16)
    else:
        gen = np.full(n, generation, dtype=np.int16)
    cols_ = {"generation": gen, "game_idx": s["game_idx"].astype(np.int16), "move_idx": s["move_idx"].astype(np.int16),
             "move": s["move"].astype(np.int16), "player": s["player"].astype(np.int8)}
    df = pd.DataFrame(cols_)
    parts = [df]
    if with_features:
        parts.append(pd.DataFrame(s["x"].astype(np.int16), columns=["x_%d" % i for i in range(F)]))
    parts.append(pd.DataFrame(s["pi"].astype(np.float64), columns=["pi_%d" % i for i in range(A)]))
    parts.append(pd.DataFrame({"z": s["z"].astype(np.int64)}))
    parts.append(pd.DataFrame({"max_deepness": s["max_deepness"].astype(np.int16), "tree_size": s["tree_size"].astype(np.int32),
                               "terminal_count": s["terminal_count"].astype(np.int32),
                               "q_value": s["q_value"].astype(np.float32)}))
    df = pd.concat(parts, axis=1)
    df.set_index(["generation", "game_idx", "move_idx"], inplace=True)
    return df


class SelfPlay:
    """Reference: self_play.py:19-156.  `nn` is a dotsboxesaz_amd.nn.NeuralNetWrapper (its engine
    plays the games) or an Engine whose evaluator is already configured."""

    def __init__(self, nn, params):
        self.params = params
        self.engine = getattr(nn, "engine", nn)
        self.samples = None

    def play_games_sync(self, games_idxs):
        idx = np.asarray(list(games_idxs), dtype=np.int64)
        if len(idx) == 0:
            return
        if not np.array_equal(idx, np.arange(idx[0], idx[0] + len(idx))):
            raise ValueError("game indices must be a contiguous range")
        self.engine.selfplay_start(len(idx), int(idx[0]))
        self.engine.run()
        got = self.engine.fetch_samples()
        self.samples = got if self.samples is None else {k: np.concatenate([self.samples[k], got[k]]) for k in got}

    async def play_games(self, game_state, games_idxs, show_progress=False):
        self.play_games_sync(games_idxs)

    def get_datasets(self, generation, with_features=True):
        e = self.engine
        return samples_to_dataframe(self.samples, generation, e.rows, e.cols, with_features)


def write_dataset(file_name, key, df):
    """utils.write_to_hdf (utils/utils.py:94-96): append to an HDFStore table; parquet when the
    file name says so (pytables is not part of this image)."""
    if str(file_name).endswith(".parquet"):
        df.reset_index().to_parquet(file_name)
        return
    import pandas as pd
    with pd.HDFStore(file_name, mode="a") as store:
        store.append(key, df, format="table")


def generate_games(hdf_file_name, generation, nn_class, n_games, params, n_workers=None, games_per_workers=10,
                   rows=None, cols=None, n_slots=None, device=0, dist=None, nn_precision=None, async_searches=False):
    """Reference: self_play.generate_games (self_play.py:291-306) called from coach.selfplay
    (coach.py:27-29).  Plays n_games with generation-1's weights (random init for generation 0,
    self_play.py:187-190) and appends the samples (+ `training` = 0) to key "fresh".
    With torch.distributed initialised (one process per GPU) the game indices are sharded over
    the ranks (the reference's np.array_split over pool workers, :294); every rank keeps its
    finished rows on the device, the packed rows are all-gathered (RCCL; host-staged under gloo)
    and EVERY rank builds the DataFrame of all n_games games -- the reference's workers all append
    to the one HDF file (self_play.py:264-265); here rank 0 writes it.
    nn_precision: None = the engine's default for the network (ResNetZero: 1, the f16x3 mode every published number of this
    repository is measured in; 0 = exact f32 MFMA, 2.6x slower).
    async_searches: honour params.self_play.mcts.max_async_searches (see engine_kwargs_from_params; default: sequential search
    per game, batching across games)."""
    from .engine import Engine
    game = _get(params, "game")
    if rows is None:
        dims = _get(game, "dims") or getattr(_get(game, "clazz"), "BOARD_DIM", (3, 3))
        rows, cols = int(dims[0]), int(dims[1])
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    first, count = shard_games(n_games, world, rank)
    n_slots = n_slots or max(1, min(count, 8192))
    model = nn_class(params)
    if generation != 0:
        model.load_parameters(generation - 1)
    # Philox streams are keyed by (seed, game, ply): sharding does not change a game
    eng = Engine(rows, cols, n_slots, evaluator=model.kind, device=device, seed=generation * 1000003,
                 nn_precision=nn_precision, **engine_kwargs_from_params(params, async_searches))
    try:
        if model.kind in ("resnet", "simplenn"):
            eng.load_state_dict(model.state_dict(), model.kind, **model.shape)
        packed = collect_rows_device(eng, count, first)
        if dist is not None and world > 1:
            packed, _ = all_gather_rows(packed, dist)
        samples = unpack_rows(packed.cpu().numpy(), eng.F, eng.A)
    finally:
        eng.close()
    df = samples_to_dataframe(samples, generation, rows, cols, True)
    df["training"] = np.zeros(len(df.index), dtype=np.int8)
    if hdf_file_name is not None and rank == 0:
        write_dataset(hdf_file_name, "fresh", df)
    return df


# ---------------------------------------------------------------------------------------
# multi-GPU: replay all-gather over RCCL (replaces the HDF-append-under-lock exchange,
# self_play.py:264-265).  Fixed-stride packed rows (DESIGN.md "replay row").
# ---------------------------------------------------------------------------------------
class _DevBuf:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}


# RowMeta of csrc/common.h (28 bytes); a packed replay row is RowMeta | x int16[3HW] | visits int32[A], padded to 8 B
ROW_META = np.dtype([("game_idx", "<i4"), ("move_idx", "<i2"), ("move", "<i2"), ("played", "<i2"),
                     ("max_deepness", "<i2"), ("tree_size", "<i4"), ("terminal_count", "<i4"), ("q_value", "<f4"),
                     ("player", "i1"), ("z", "i1"), ("pad", "<i2")])
assert ROW_META.itemsize == 28


def row_bytes(F, A):
    return (ROW_META.itemsize + 2 * F + 4 * A + 7) // 8 * 8


def unpack_rows(packed, F, A):
    """Packed replay rows (uint8 [n, row_bytes], any rank order) -> the sample dict of
    Engine.fetch_samples(), sorted by (game_idx, move_idx); pi = visits / (sum or 1.0) in float64
    (self_play.py:114-115)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n = packed.shape[0]
    if n and packed.shape[1] != row_bytes(F, A):
        raise ValueError("row stride %d does not match a %d-feature / %d-action board" % (packed.shape[1], F, A))
    m0 = ROW_META.itemsize
    meta = np.ascontiguousarray(packed[:, :m0]).view(ROW_META).reshape(n)
    x = np.ascontiguousarray(packed[:, m0:m0 + 2 * F]).view("<i2").reshape(n, F)
    vis = np.ascontiguousarray(packed[:, m0 + 2 * F:m0 + 2 * F + 4 * A]).view("<i4").reshape(n, A)
    vs = vis.sum(axis=1, dtype=np.int64).astype(np.float64)
    pi = vis.astype(np.float64) / np.where(vs == 0, 1.0, vs)[:, None]
    out = dict(game_idx=meta["game_idx"].astype(np.int32), move_idx=meta["move_idx"].astype(np.int16),
               move=meta["move"].astype(np.int16), player=meta["player"].astype(np.int8), x=x.astype(np.int16),
               visits=vis.astype(np.int32), pi=pi, z=meta["z"].astype(np.int8),
               max_deepness=meta["max_deepness"].astype(np.int16), tree_size=meta["tree_size"].astype(np.int32),
               terminal_count=meta["terminal_count"].astype(np.int32), q_value=meta["q_value"].astype(np.float32),
               played=meta["played"].astype(np.int16))
    order = np.lexsort((out["move_idx"], out["game_idx"]))
    return {k: v[order] for k, v in out.items()}


def _warn_pool_resets(engine, counters):
    """The reference's search trees are unbounded; a slot's node pool is not.  When the subtree kept by tree reuse would not leave
    room for the next search, the engine starts that move from a fresh root (dbaz_counters.pool_resets) -- say so once per run."""
    n = int(counters.get("pool_resets", 0))
    if n:
        import warnings
        warnings.warn("%d moves (of %d) were searched from a fresh root because the reused subtree did not leave mcts_num_read + 2 nodes "
                      "of the slot's pool free: raise nodes_per_slot (now %d) to keep the reference's tree reuse on every move"
                      % (n, int(counters.get("moves_played", 0)), int(getattr(engine, "nodes_per_slot", 0) or engine.cfg.nodes_per_slot)))


def collect_rows_device(engine, count, first):
    """Play games first .. first+count-1 and return their packed rows as ONE uint8 CUDA tensor
    [n, row_bytes]; the rows never visit the host.  Whenever the device row buffer fills (the
    engine's backpressure: finished games wait in PH_EMIT) its content is cloned on the device and
    the buffer emptied."""
    import torch
    dev = torch.device("cuda", engine.cfg.device)
    chunks = []
    if count > 0:
        engine.selfplay_start(count, first)
        while True:
            engine._ck(engine._L.dbaz_run(engine.h, 0))
            ptr, n, rb = engine.replay_rows_dev()
            if n:
                chunks.append(torch.as_tensor(_DevBuf(ptr, n * rb), device=dev).view(n, rb).clone())
            c = engine.counters()
            engine.replay_rows_clear()
            if c["active_slots"] == 0:
                break
        _warn_pool_resets(engine, c)
    if chunks:
        return torch.cat(chunks, dim=0)
    return torch.zeros((0, engine.row_bytes), dtype=torch.uint8, device=dev)


def all_gather_rows(rows, dist):
    """rows: uint8 tensor [n_i, row_bytes] (any device).  Returns (uint8 [sum n_i, row_bytes], counts),
    rank order.  Under RCCL ("nccl") the exchange runs device to device; a gloo group (CPU tests,
    two ranks sharing one GPU) is staged through the host and the result returned on rows' device."""
    import torch
    world = dist.get_world_size()
    home = rows.device
    if rows.is_cuda and dist.get_backend() == "gloo":
        rows = rows.cpu()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    padded = torch.zeros((mx, rows.shape[1]), dtype=torch.uint8, device=rows.device)
    padded[: rows.shape[0]] = rows
    out = torch.empty((world * mx, rows.shape[1]), dtype=torch.uint8, device=rows.device)
    dist.all_gather_into_tensor(out, padded)
    keep = torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)
    return keep.to(home), counts


def gather_replay(engine, dist, synthetic_rows=0):
    """All-gather the finished rows that sit in HBM (dbaz_replay_rows_dev).  Returns
    (total rows, milliseconds).  synthetic_rows > 0 pads every rank's shard to that many rows
    (benchmarking the exchange before games have finished)."""
    import torch
    ptr, n, rb = engine.replay_rows_dev()
    dev = torch.device("cuda", engine.cfg.device)
    if n > 0:
        rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=dev).view(n, rb)
    else:
        rows = torch.zeros((0, rb), dtype=torch.uint8, device=dev)
    if synthetic_rows > n:
        rows = torch.cat([rows, torch.zeros((synthetic_rows - n, rb), dtype=torch.uint8, device=dev)], dim=0)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    allrows, counts = all_gather_rows(rows, dist)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    return int(sum(counts)), ms


# ---------------------------------------------------------------------------------------
# two-model match play + Elo (self_play.compute_elo, self_play.py:309-344)
# ---------------------------------------------------------------------------------------
def elo_rating2(elo0, elo1, n0, n1, K=30):
    """utils/utils.py:124-132."""
    import math
    p1 = 1.0 / (1 + 1.0 * math.pow(10, (elo0 - elo1) / 400))
    p0 = 1 - p1
    return elo0 + K * (n0 * p1 - n1 * p0), elo1 + K * (n1 * p0 - n0 * p1)


def match_winners(samples, generations):
    """(n0, n1): wins of generations[0] / generations[1] counted as the reference does
    (self_play.py:336-338): per game the first row (lowest move_idx) with z == 1 names the winner;
    drawn games have no such row.  Seats: model = player XOR (game_idx & 1)."""
    gen = np.where((samples["player"] ^ (samples["game_idx"] & 1)) == 0, generations[0], generations[1])
    n0 = n1 = 0
    for g in np.unique(samples["game_idx"]):
        r = np.nonzero((samples["game_idx"] == g) & (samples["z"] == 1))[0]
        if len(r) == 0:
            continue
        first = r[np.argmin(samples["move_idx"][r])]
        if gen[first] == generations[0]:
            n0 += 1
        else:
            n1 += 1
    return n0, n1


def compute_elo(elo_params, params, generations, elos, nn_classes=None, rows=None, cols=None, n_slots=None, device=0,
                nn_precision=None):
    """Reference: self_play.compute_elo(elo_params, [params0, params1], [gen0, gen1], (elo0, elo1)).
    The two models play elo_params.n_games games against each other on the GPU: the model of the
    player to move at the root runs that move's whole search (self_play.py:59,237-239), seats are
    swapped on odd games, the `self_play_override` of elo_params applies (no tree reuse, no noise,
    1200 reads in the shipped configuration, configuration.py:107-113).
    nn_precision: as in generate_games (None = f16x3 for ResNetZero).
    Returns (elo0, elo1, n1 / number of decided games)."""
    from .engine import Engine
    p0, p1 = params
    over = _get(elo_params, "self_play_override", {}) or {}
    kw = engine_kwargs_from_params(p0)
    if "reuse_mcts_tree" in over:
        kw["reuse_tree"] = bool(_get(over, "reuse_mcts_tree"))
    if "noise" in over:
        nz = _get(over, "noise")
        kw["noise"] = (float(nz[0]), float(nz[1]))
    m = _get(over, "mcts")
    if m is not None and _get(m, "mcts_num_read") is not None:
        kw["mcts_num_read"] = int(_get(m, "mcts_num_read"))
    if rows is None:
        game = _get(p0, "game")
        dims = _get(game, "dims") or getattr(_get(game, "clazz"), "BOARD_DIM", (3, 3))
        rows, cols = int(dims[0]), int(dims[1])
    n_games = int(_get(elo_params, "n_games"))
    classes = nn_classes or [_get(_get(p, "nn"), "model_class") for p in params]
    models = [classes[0](p0), classes[1](p1)]
    for mdl, gen in zip(models, generations):
        if gen != 0:
            mdl.load_parameters(gen)  # compare_models: generation g itself (self_play.py:190)
    eng = Engine(rows, cols, n_slots or max(1, min(n_games, 4096)), evaluator=models[0].kind, evaluator2=models[1].kind,
                 match_play=True, device=device, nn_precision=nn_precision, **kw)
    try:
        for i, mdl in enumerate(models):
            eng.load_state_dict(mdl.state_dict(), mdl.kind, model=i, **mdl.shape)
        eng.selfplay_start(n_games, 0)
        eng.run()
        got = eng.fetch_samples()
    finally:
        eng.close()
    n0, n1 = match_winners(got, generations)
    e0, e1 = elo_rating2(elos[0], elos[1], n0, n1, K=30)
    decided = n0 + n1
    return e0, e1, (n1 / decided if decided else float("nan"))

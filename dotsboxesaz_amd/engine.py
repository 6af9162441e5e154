"""numpy-level wrapper of the HIP engine handle (one handle per GPU / rank)."""
import ctypes as C

import numpy as np

from . import _lib

_p = lambda a: a.ctypes.data if a is not None else None  # noqa: E731


def _temperature_items(temperature):
    temperature = {0: 1.0, 12: 0.02} if temperature is None else temperature
    items = sorted((int(k), float(v)) for k, v in dict(temperature).items())
    if len(items) > 8:
        raise ValueError("at most 8 temperature schedule entries")
    return items


def _dense_from_grouped(w, channels):
    """Conv2d(groups=g) weight [C][C/g][kh][kw] (the reference's n_groups, nn.py:61-71) -> the dense [C][C][kh][kw] weight of the
    same function: output group i sees input group i only, every other product is an exact zero."""
    cpg = w.shape[1]
    if channels % cpg:
        raise ValueError("grouped conv weight %s does not divide %d channels" % (w.shape, channels))
    dense = np.zeros((channels, channels) + w.shape[2:], np.float32)
    for g in range(channels // cpg):
        dense[g * cpg:(g + 1) * cpg, g * cpg:(g + 1) * cpg] = w[g * cpg:(g + 1) * cpg]
    return dense


class Engine:
    """Self-play rollout engine on one MI355X.

    Parameters mirror the reference's configuration (configuration.py:82-100):
    mcts_num_read, cpuct=(c, base), noise=(alpha, coeff), temperature={ply: T},
    reuse_tree.  `evaluator` is one of "formula", "uniform", "resnet", "simplenn",
    "external".  nn_precision: None = 1 (f16x3 split MFMA, f32-grade) for "resnet", 0 (exact f32 MFMA) otherwise.
    """

    EVALUATORS = {"formula": _lib.EVAL_FORMULA_HASH, "uniform": _lib.EVAL_FORMULA_UNIFORM,
                  "resnet": _lib.EVAL_RESNET, "simplenn": _lib.EVAL_SIMPLENN, "external": _lib.EVAL_EXTERNAL}

    def __init__(self, rows, cols, n_slots, mcts_num_read=800, cpuct=(1.25, 19652), noise=(0.0, 0.0),
                 temperature=None, reuse_tree=True, evaluator="formula", nodes_per_slot=0, seed=0, device=0,
                 max_out_rows=0, nn_precision=None, match_play=False, evaluator2="formula", transposition_cache=True,
                 max_pending_evals=1, selfplay_pending=False, eval_round=0, eval_defer_max=0, debug_flags=0):
        self._L = _lib.load()
        self.rows, self.cols = int(rows), int(cols)
        self.H, self.W = self.rows + 1, self.cols + 1
        self.A = 2 * self.H * self.W
        self.F = 3 * self.H * self.W
        self.E = 2 * self.rows * self.cols + self.rows + self.cols
        self.n_slots = int(n_slots)
        cfg = _lib.Config()
        cfg.rows, cfg.cols, cfg.n_slots, cfg.nodes_per_slot = self.rows, self.cols, self.n_slots, int(nodes_per_slot)
        cfg.mcts_num_read = int(mcts_num_read)
        cfg.cpuct, cfg.cpuct_base = float(cpuct[0]), float(cpuct[1])
        cfg.noise_alpha, cfg.noise_coeff = float(noise[0]), float(noise[1])
        cfg.reuse_tree = int(bool(reuse_tree))
        items = _temperature_items(temperature)
        cfg.n_temp = len(items)
        for i, (k, v) in enumerate(items):
            cfg.temp_idx[i], cfg.temp_val[i] = k, v
        cfg.evaluator = self.EVALUATORS[evaluator] if isinstance(evaluator, str) else int(evaluator)
        cfg.device, cfg.seed, cfg.max_out_rows = int(device), int(seed), int(max_out_rows)
        # None: the advertised mode of the network -- f16x3 (f32-grade, exact-f32 safety net on the device) for ResNetZero, exact f32
        # for SimpleNN (whose f16x3 path has no safety net) and the formula evaluators
        if nn_precision is None:
            nn_precision = 1 if cfg.evaluator == _lib.EVAL_RESNET else 0
        cfg.nn_precision = int(nn_precision)
        cfg.match_play = int(bool(match_play))
        cfg.evaluator2 = self.EVALUATORS[evaluator2] if isinstance(evaluator2, str) else int(evaluator2)
        # True: on for network evaluators; False: off; "force": on for the formula evaluators too (parity tests)
        cfg.transposition_cache = 2 if transposition_cache == "force" else (0 if transposition_cache else 1)
        cfg.max_pending_evals = int(max_pending_evals)
        # self-play searches in waves of max_pending_evals simulations per game (self_play.py:27-30's max_async_searches)
        cfg.selfplay_pending = int(bool(selfplay_pending))
        # "full rounds only": 0 = the network's own round, -1 = off, r > 0 = rounds of r leaves (tests)
        cfg.eval_round, cfg.eval_defer_max = int(eval_round), int(eval_defer_max)
        cfg.debug_flags = int(debug_flags)
        self.cfg = cfg
        self._drained = []
        self.h = C.c_void_p()
        rc = self._L.dbaz_create(C.byref(cfg), C.byref(self.h))
        if rc != _lib.OK:
            _lib.check(None, rc)
        self.nodes_per_slot = int(self._L.dbaz_nodes_per_slot(self.h))  # the pool size in effect (0 asked for the default rule)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self._L.dbaz_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(self.h, rc)

    # ---------------------------------------------------------------- rules (G1-G6)
    def rules_init(self, n):
        st = dict(edges=np.zeros((n, 4), np.uint64), b2c2=np.zeros((n, 2), np.int16),
                  to_play=np.zeros(n, np.int8), just_played=np.zeros(n, np.int8))
        self._ck(self._L.dbaz_rules_init(self.h, n, _p(st["edges"]), _p(st["b2c2"]), _p(st["to_play"]),
                                         _p(st["just_played"])))
        return st

    def rules_valid_moves(self, st):
        n = len(st["to_play"])
        out = np.zeros((n, self.A), np.uint8)
        self._ck(self._L.dbaz_rules_valid_moves(self.h, n, _p(st["edges"]), _p(out)))
        return out.astype(bool)

    def rules_play(self, st, moves):
        """In-place play_ on every state; returns (n_closed int8[n], closed_lc int8[n,4]).
        Raises ValueError if any move is illegal (legal ones are still applied)."""
        n = len(st["to_play"])
        moves = np.ascontiguousarray(moves, np.int32)
        nc = np.zeros(n, np.int8)
        lc = np.zeros((n, 4), np.int8)
        self._ck(self._L.dbaz_rules_play(self.h, n, _p(st["edges"]), _p(st["b2c2"]), _p(st["to_play"]),
                                         _p(st["just_played"]), _p(moves), _p(nc), _p(lc)))
        return nc, lc

    def rules_play_status(self, st, moves):
        """Like rules_play but returns n_closed (-1 = illegal) instead of raising."""
        n = len(st["to_play"])
        moves = np.ascontiguousarray(moves, np.int32)
        nc = np.zeros(n, np.int8)
        lc = np.zeros((n, 4), np.int8)
        rc = self._L.dbaz_rules_play(self.h, n, _p(st["edges"]), _p(st["b2c2"]), _p(st["to_play"]),
                                     _p(st["just_played"]), _p(moves), _p(nc), _p(lc))
        if rc not in (_lib.OK, _lib.EILLEGAL):
            self._ck(rc)
        return nc, lc

    def rules_result(self, st):
        n = len(st["to_play"])
        out = np.zeros(n, np.int8)
        self._ck(self._L.dbaz_rules_result(self.h, n, _p(st["b2c2"]), _p(st["to_play"]), _p(out)))
        return out

    def rules_features(self, st):
        n = len(st["to_play"])
        out = np.zeros((n, 3, self.H, self.W), np.int16)
        self._ck(self._L.dbaz_rules_features(self.h, n, _p(st["edges"]), _p(st["b2c2"]), _p(st["to_play"]), _p(out)))
        return out

    # ---------------------------------------------------------------- network (N1-N3)
    def load_state_dict(self, state_dict, kind="resnet", channels=64, blocks=20, head_channels=16, value_fc=8, model=0):
        """state_dict: mapping of the reference's key names to arrays (torch tensors or numpy).
        model: 0 or 1 (the two players of match play)."""
        self._ck(self._L.dbaz_nn_select_model(self.h, int(model)))
        try:
            self._load_state_dict(state_dict, kind, channels, blocks, head_channels, value_fc)
        finally:
            self._L.dbaz_nn_select_model(self.h, 0)

    def _load_state_dict(self, state_dict, kind, channels, blocks, head_channels, value_fc):
        self._ck(self._L.dbaz_nn_configure(self.h, self.EVALUATORS[kind], channels, blocks, head_channels, value_fc))
        for k, v in state_dict.items():
            if k.endswith("num_batches_tracked"):
                continue
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            a = np.ascontiguousarray(a, np.float32)
            if kind == "resnet" and ".resblocks." in k and k.endswith(".weight") and a.ndim == 4 and a.shape[0] == channels \
                    and a.shape[1] < channels:
                a = _dense_from_grouped(a, channels)
            self._ck(self._L.dbaz_nn_set_tensor(self.h, k.encode(), _p(a), a.size))
        self._ck(self._L.dbaz_nn_commit(self.h))

    def predict(self, X):
        """NeuralNetWrapper.predict_sync: X [n,3,H,W] -> (p [n,A], v [n,1]) float32."""
        X = np.ascontiguousarray(X, np.float32)
        n = X.shape[0]
        assert X.shape[1:] == (3, self.H, self.W), X.shape
        p = np.zeros((n, self.A), np.float32)
        v = np.zeros((n, 1), np.float32)
        self._ck(self._L.dbaz_nn_predict(self.h, n, _p(X), _p(p), _p(v)))
        return p, v

    # ---------------------------------------------------------------- search (M1-M9)
    def set_search_params(self, cpuct=(1.25, 19652), dirichlet=(0.0, 0.0)):
        self._ck(self._L.dbaz_set_search_params(self.h, float(cpuct[0]), float(cpuct[1]), float(dirichlet[0]),
                                                float(dirichlet[1])))

    def set_positions(self, move_lists=None):
        if move_lists is None:
            self._ck(self._L.dbaz_set_positions(self.h, None, None))
            return
        assert len(move_lists) == self.n_slots
        off = np.zeros(self.n_slots + 1, np.int32)
        off[1:] = np.cumsum([len(m) for m in move_lists])
        flat = np.ascontiguousarray(np.concatenate([np.asarray(m, np.int16).ravel() for m in move_lists] +
                                                   [np.zeros(1, np.int16)]), np.int16)
        self._ck(self._L.dbaz_set_positions(self.h, _p(flat), _p(off)))

    def _search_args(self, num_reads, noise):
        nr = None
        if num_reads is not None:
            nr = np.ascontiguousarray(np.broadcast_to(np.asarray(num_reads, np.int32), (self.n_slots,)), np.int32)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, np.float64)
            assert nz.shape == (self.n_slots, self.A)
        return nr, nz

    def search(self, num_reads=None, noise=None):
        nr, nz = self._search_args(num_reads, noise)
        self._ck(self._L.dbaz_search(self.h, _p(nr), _p(nz)))

    def set_pending(self, k, virtual_visits=True):
        """max_pending_evals of the following searches (1..the handle's max_pending_evals).  virtual_visits=False: the
        reference's bookkeeping (visits added at backup); True: also counted on the path at selection time."""
        self._ck(self._L.dbaz_set_pending(self.h, int(k), int(bool(virtual_visits))))

    def search_timed(self, time_limit, num_reads=None, noise=None):
        """UCT_search with its wall-clock cut-off (seconds; None / 0 = the reference's 120 s default)."""
        nr, nz = self._search_args(num_reads, noise)
        self._ck(self._L.dbaz_search_timed(self.h, _p(nr), _p(nz), float(time_limit or 0.0)))

    def search_external(self, evaluate, num_reads=None, noise=None):
        """UCT_search with a host evaluator: evaluate(x int16 [m,3,H,W]) -> (p [m,A], v [m])."""
        nr, nz = self._search_args(num_reads, noise)
        self._ck(self._L.dbaz_search_begin(self.h, _p(nr), _p(nz)))
        x = np.zeros((self.n_slots, 3, self.H, self.W), np.int16)
        need = np.zeros(self.n_slots, np.uint8)
        na = C.c_int32()
        P = np.zeros((self.n_slots, self.A), np.float32)
        V = np.zeros(self.n_slots, np.float32)
        while True:
            self._ck(self._L.dbaz_select(self.h, C.byref(na), _p(x), _p(need)))
            if na.value == 0:
                break
            idx = np.nonzero(need)[0]
            if len(idx):
                p, v = evaluate(x[idx])
                P[idx] = np.asarray(p, np.float32)
                V[idx] = np.asarray(v, np.float32).reshape(-1)
            self._ck(self._L.dbaz_expand_backup(self.h, _p(P), _p(V)))

    def roots(self):
        n, A = self.n_slots, self.A
        out = dict(priors=np.zeros((n, A), np.float64), total_value=np.zeros((n, A), np.float32),
                   visits=np.zeros((n, A), np.int32), changed=np.zeros((n, A), np.int32),
                   stats=np.zeros((n, 3), np.int32), q=np.zeros(n, np.float32), root_tv=np.zeros(n, np.float32),
                   root_nv=np.zeros(n, np.int32))
        self._ck(self._L.dbaz_get_roots(self.h, _p(out["priors"]), _p(out["total_value"]), _p(out["visits"]),
                                        _p(out["changed"]), _p(out["stats"]), _p(out["q"]), _p(out["root_tv"]),
                                        _p(out["root_nv"])))
        return out

    def root_states(self):
        n = self.n_slots
        out = dict(edges=np.zeros((n, 4), np.uint64), b2c2=np.zeros((n, 2), np.int16), to_play=np.zeros(n, np.int8),
                   just_played=np.zeros(n, np.int8), result=np.zeros(n, np.int8), expanded=np.zeros(n, np.int8))
        self._ck(self._L.dbaz_get_root_states(self.h, _p(out["edges"]), _p(out["b2c2"]), _p(out["to_play"]),
                                              _p(out["just_played"]), _p(out["result"]), _p(out["expanded"])))
        return out

    def advance(self, moves, reuse_tree=True):
        mv = np.ascontiguousarray(np.broadcast_to(np.asarray(moves, np.int32), (self.n_slots,)), np.int32)
        self._ck(self._L.dbaz_advance(self.h, _p(mv), int(bool(reuse_tree))))

    # ---------------------------------------------------------------- self-play (D1-D3)
    def selfplay_script(self, game_idx, moves, noise=None):
        mv = np.ascontiguousarray(moves, np.int16)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, np.float64)
            assert nz.shape == (len(mv), self.A)
        self._ck(self._L.dbaz_selfplay_script(self.h, int(game_idx), _p(mv), len(mv), _p(nz)))

    def selfplay_fastforward(self, plies):
        pl = np.ascontiguousarray(plies, np.int32)
        assert pl.shape == (self.n_slots,)
        self._ck(self._L.dbaz_selfplay_fastforward(self.h, _p(pl)))

    def selfplay_stagger(self, first_reads):
        """Benchmark population: read budget of each slot's first search (0 = the driver rule)."""
        fr = np.ascontiguousarray(first_reads, np.int32)
        assert fr.shape == (self.n_slots,)
        self._ck(self._L.dbaz_selfplay_stagger(self.h, _p(fr)))

    def selfplay_quickplay(self, plies, reads):
        """Benchmark population: the first plies[i] plies of slot i's first game use `reads` simulations per move."""
        pl = np.ascontiguousarray(plies, np.int32)
        assert pl.shape == (self.n_slots,)
        self._ck(self._L.dbaz_selfplay_quickplay(self.h, _p(pl), int(reads)))

    def selfplay_start(self, n_games, first_game_idx=0):
        self._ck(self._L.dbaz_selfplay_start(self.h, int(n_games), int(first_game_idx)))

    def step(self, k=1):
        self._ck(self._L.dbaz_step(self.h, int(k)))

    def run(self, max_steps=0):
        """Play until every game is finished.  Finished rows that do not fit the device
        output buffer are drained to the host on the way; fetch_samples() returns them all."""
        while True:
            self._ck(self._L.dbaz_run(self.h, int(max_steps)))
            c = self.counters()
            if max_steps or c["active_slots"] == 0 or c["blocked_slots"] < c["active_slots"]:
                self._warn_pool_resets(c)
                return
            self._drained.append(self._fetch_once())

    def _warn_pool_resets(self, c):
        """The reference's trees are unbounded; a slot's node pool is not: a move whose reused subtree would not leave
        mcts_num_read + 2 nodes free starts from a fresh root instead (counters()["pool_resets"]).  Said once per handle."""
        n = int(c.get("pool_resets", 0))
        if n and not getattr(self, "_warned_resets", False):
            import warnings
            self._warned_resets = True
            warnings.warn("%d of %d moves were searched from a fresh root because the reused subtree did not leave mcts_num_read + 2 "
                          "nodes of the slot's pool free: raise nodes_per_slot (now %d) to keep the reference's tree reuse on every move"
                          % (n, int(c.get("moves_played", 0)), self.nodes_per_slot))

    def sync(self):
        self._ck(self._L.dbaz_sync(self.h))

    def timing_begin(self):
        self._ck(self._L.dbaz_timing_begin(self.h))

    def timing_end(self):
        self._ck(self._L.dbaz_timing_end(self.h))

    def counters(self):
        c = _lib.Counters()
        self._ck(self._L.dbaz_get_counters(self.h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in _lib.Counters._fields_}

    def fetch_samples(self):
        """Rows of SelfPlay.get_datasets for the games finished so far (drains the buffer),
        sorted by (game_idx, move_idx)."""
        parts = self._drained + [self._fetch_once()]
        self._drained = []
        out = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
        order = np.lexsort((out["move_idx"], out["game_idx"]))
        return {k: v[order] for k, v in out.items()}

    def _fetch_once(self):
        n = C.c_int32()
        self._ck(self._L.dbaz_fetch_samples(self.h, 0, C.byref(n), *([None] * 13)))
        m = n.value
        out = dict(game_idx=np.zeros(m, np.int32), move_idx=np.zeros(m, np.int16), move=np.zeros(m, np.int16),
                   player=np.zeros(m, np.int8), x=np.zeros((m, self.F), np.int16), visits=np.zeros((m, self.A), np.int32),
                   pi=np.zeros((m, self.A), np.float64), z=np.zeros(m, np.int8), max_deepness=np.zeros(m, np.int16),
                   tree_size=np.zeros(m, np.int32), terminal_count=np.zeros(m, np.int32),
                   q_value=np.zeros(m, np.float32), played=np.zeros(m, np.int16))
        if m:
            self._ck(self._L.dbaz_fetch_samples(
                self.h, m, C.byref(n), _p(out["game_idx"]), _p(out["move_idx"]), _p(out["move"]), _p(out["player"]),
                _p(out["x"]), _p(out["visits"]), _p(out["pi"]), _p(out["z"]), _p(out["max_deepness"]),
                _p(out["tree_size"]), _p(out["terminal_count"]), _p(out["q_value"]), _p(out["played"])))
        return out

    def replay_rows_dev(self):
        """(device pointer, n_rows, row_bytes) of the packed finished rows (for the RCCL all-gather)."""
        ptr, n, rb = C.c_void_p(), C.c_int32(), C.c_int32()
        self._ck(self._L.dbaz_replay_rows_dev(self.h, C.byref(ptr), C.byref(n), C.byref(rb)))
        return ptr.value, n.value, rb.value

    # ---- training data path: replay rows in HBM -> dataset -> batches in HBM
    @property
    def row_bytes(self):
        return (28 + self.F * 2 + self.A * 4 + 7) // 8 * 8

    def replay_rows_clear(self):
        """Empty the finished-row buffer (the caller keeps the rows it took with replay_rows_dev on the device)."""
        self._ck(self._L.dbaz_replay_rows_clear(self.h))

    def dataset_select(self, which):
        """Address dataset `which` (0..3) with the following dataset_* calls."""
        self._ck(self._L.dbaz_dataset_select(self.h, C.c_int32(which)))
        self._ds_cur = int(which)

    def dataset_begin(self):
        self._ck(self._L.dbaz_dataset_begin(self.h))

    def dataset_add_rows(self, rows, sel=None):
        """rows: (device pointer, n_rows, row_bytes) as returned by replay_rows_dev(), or any object
        with data_ptr()/shape (a torch uint8 CUDA tensor [n, row_bytes], e.g. the all-gathered
        replay).  sel: host int32 row indices in dataset order (None = all rows)."""
        if isinstance(rows, tuple):
            ptr, n, rb = rows
        else:
            if not rows.is_cuda or not rows.is_contiguous():
                raise ValueError("replay rows must be a contiguous CUDA tensor")
            ptr, n, rb = rows.data_ptr(), int(rows.shape[0]), int(rows.shape[1])
        s, ns = None, 0
        if sel is not None:
            s = np.ascontiguousarray(sel, dtype=np.int32)
            ns = len(s)
        self._ck(self._L.dbaz_dataset_add_rows(self.h, C.c_void_p(ptr), C.c_int64(n), C.c_int32(rb), C.c_void_p(_p(s)),
                                               C.c_int64(ns)))

    def dataset_finish(self, pos_average=False, order=None):
        """order: dataset order as a permutation of the staged rows (None = staging order)."""
        n = C.c_int64()
        o = np.ascontiguousarray(order, dtype=np.int32) if order is not None else None
        self._ck(self._L.dbaz_dataset_finish(self.h, C.c_int32(1 if pos_average else 0), C.c_void_p(_p(o)), C.byref(n)))
        self._ds_sizes = getattr(self, "_ds_sizes", {})
        self._ds_sizes[getattr(self, "_ds_cur", 0)] = n.value
        return n.value

    def dataset_fetch(self):
        """Host copies (features int16 [n,3HW], policy float32 [n,A], value float32 [n])."""
        n = getattr(self, "_ds_sizes", {}).get(getattr(self, "_ds_cur", 0), 0)
        x = np.empty((n, self.F), dtype=np.int16)
        pi = np.empty((n, self.A), dtype=np.float32)
        z = np.empty(n, dtype=np.float32)
        self._ck(self._L.dbaz_dataset_fetch(self.h, C.c_void_p(_p(x)), C.c_void_p(_p(pi)), C.c_void_p(_p(z))))
        return x, pi, z

    def dataset_batch(self, idx, sym=0):
        """Rows idx of the dataset under symmetry sym as torch CUDA tensors
        (boards float32 [n,3,H,W], pi [n,A], z [n,1]) -- written by the HIP kernel, no host copy; asynchronous, ordered on torch's
        current stream like any torch op."""
        import torch
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        n = len(idx)
        dev = torch.device("cuda", self.cfg.device)
        boards = torch.empty((n, 3, self.H, self.W), dtype=torch.float32, device=dev)
        pi = torch.empty((n, self.A), dtype=torch.float32, device=dev)
        z = torch.empty((n, 1), dtype=torch.float32, device=dev)
        # queued on torch's CURRENT stream: the tensors above come from torch's stream-ordered caching allocator, so a kernel on any
        # other stream could overwrite a recycled block while work queued on its previous owner is still pending
        with torch.cuda.device(dev):
            self._ck(self._L.dbaz_dataset_batch_on(self.h, C.c_void_p(_p(idx)), C.c_int32(n), C.c_int32(sym), C.c_void_p(boards.data_ptr()),
                                                   C.c_void_p(pi.data_ptr()), C.c_void_p(z.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        return boards, pi, z

    def symmetry_apply(self, sym, boards=None, policies=None):
        """SymmetriesGenerator transform `sym` of torch CUDA float32 tensors (out of place)."""
        import torch
        n = int((boards if boards is not None else policies).shape[0])
        for t in (boards, policies):
            if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
                raise ValueError("symmetry_apply needs contiguous float32 CUDA tensors")
        torch.cuda.current_stream(boards.device if boards is not None else policies.device).synchronize()
        bo = torch.empty_like(boards) if boards is not None else None
        po = torch.empty_like(policies) if policies is not None else None
        dp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        self._ck(self._L.dbaz_symmetry_apply(self.h, C.c_int32(sym), dp(boards), dp(policies), C.c_int64(n), dp(bo), dp(po)))
        return bo, po


def symmetry_table(rows, cols, sym):
    """src[a'] with out[a'] = in[src[a']] over the two edge planes (host only, no GPU needed)."""
    L = _lib.load()
    lut = np.empty(2 * (rows + 1) * (cols + 1), dtype=np.int32)
    _lib.check(None, L.dbaz_symmetry_table(C.c_int32(rows), C.c_int32(cols), C.c_int32(sym), C.c_void_p(lut.ctypes.data)))
    return lut

"""Generation loop with the replay kept in HBM: the data flow of the reference's
`coach.learn_to_play` (coach.py:124-161) -- selfplay -> train_nn -> compute_elo -> next generation --
with the HDF file between the stages replaced by packed replay rows that never leave the GPU:

    selfplay(g)   HIP engine plays params.self_play.num_games games with generation g-1's weights
                  (self_play.py:187-190); the finished rows stay on the device (all-gathered over RCCL
                  when torch.distributed is initialised) and enter the ReplayStore with the
                  train/validation flags of coach.py:59-63
    train_nn(g)   window rule of coach.py:148-149, HDFStoreDataset semantics on the device
                  (train_data.ReplayDataset), optimizer step on csrc/train.hip + train_net.hip behind torch autograd (train.train)
    compute_elo   two-model match play on the device (self_play.compute_elo)

This is the thin caller of the hot path, not a port of the reference's control plane: no HDF, no
tensorboard dependency (any object with add_scalar/add_scalars serves as `writer`).
"""
import time

from . import self_play as sp
from . import train as T
from . import train_data as TD


def _get(d, k, default=None):
    if isinstance(d, dict):
        return d.get(k, default)
    return getattr(d, k, default)


class NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_scalars(self, *a, **k):
        pass

    def add_text(self, *a, **k):
        pass

    def close(self):
        pass


class Coach:
    def __init__(self, params, rows, cols, device=0, n_slots=None, dist=None, nn_precision=1):
        self.params, self.rows, self.cols = params, int(rows), int(cols)
        self.device, self.dist, self.n_slots, self.nn_precision = device, dist, n_slots, nn_precision
        self.model_class = _get(_get(params, "nn"), "model_class")
        self.engine = None
        self.store = None

    # ---- the one engine handle of this rank (tree pools sized for self-play)
    def _engine(self, kind):
        from .engine import Engine
        if self.engine is None:
            n_games = int(_get(_get(self.params, "self_play"), "num_games"))
            world = self.dist.get_world_size() if self.dist is not None else 1
            slots = self.n_slots or max(1, min(-(-n_games // world), 8192))
            self.engine = Engine(self.rows, self.cols, slots, evaluator=kind, device=self.device,
                                 nn_precision=self.nn_precision if kind == "resnet" else 0,
                                 **sp.engine_kwargs_from_params(self.params))
            self.store = TD.ReplayStore(self.engine)
        return self.engine

    def selfplay(self, generation):
        """coach.selfplay (coach.py:17-32) -> self_play.generate_games, rows kept in HBM."""
        model = self.model_class(self.params)
        if generation != 0:
            model.load_parameters(generation - 1)
        e = self._engine(model.kind)
        e.load_state_dict(model.state_dict(), model.kind, **model.shape)
        n_games = int(_get(_get(self.params, "self_play"), "num_games"))
        rank, world = (self.dist.get_rank(), self.dist.get_world_size()) if self.dist is not None else (0, 1)
        first, count = sp.shard_games(n_games, world, rank)
        tick = time.time()
        rows = sp.collect_rows_device(e, count, first)  # backpressure drains stay on the device
        if self.dist is not None and world > 1:
            rows, _ = sp.all_gather_rows(rows, self.dist)
        self.store.add_generation(generation, rows, train_split=float(_get(_get(_get(self.params, "nn"), "train_params"), "train_split", 0.9)))
        c = e.counters()  # this rank's games: network evaluations, those the exact-f32 safety net redid, tree-reuse fallbacks
        return dict(rows=int(rows.shape[0]), seconds=time.time() - tick, games=n_games, nn_evals=int(c["nn_evals"]),
                    f32_fallback_evals=int(c["f32_fallback_evals"]), pool_resets=int(c["pool_resets"]), moves_played=int(c["moves_played"]))

    def train_nn(self, generation, writer=None):
        """coach.train_nn (coach.py:35-96) on the device-resident window."""
        from .nn import NeuralNetWrapper
        nnp = _get(self.params, "nn")
        tp = _get(nnp, "train_params")
        sched = _get(tp, "lr_scheduler")
        if sched is not None:
            tp["lr"] = sched(generation)
        gmin = T.window_where(generation)
        n_samples, split, avg = int(_get(tp, "max_samples_per_gen", int(1e12))), float(_get(tp, "train_split", 0.9)), bool(_get(tp, "pos_average", False))
        model = self.model_class(self.params)
        wrapper = NeuralNetWrapper(model, self.params, engine=self.engine)
        if _get(tp, "symmetries") is None or not hasattr(_get(tp, "symmetries"), "draw"):
            tp["symmetries"] = TD.SymmetriesGenerator(self.engine)
        writer = writer or NullWriter()

        train_ds = self.store.dataset(train=True, min_generation=gmin, n_samples=int(n_samples * split), pos_average=avg)
        val_ds = self.store.dataset(train=False, min_generation=gmin, n_samples=int(n_samples * (1 - split)), pos_average=avg)
        self.store.drop_before(gmin)  # (the window's lower edge only moves forward: older rows would stay in HBM for nothing)
        return wrapper.train(train_ds, val_ds if len(val_ds) else None, writer, generation)

    def learn_to_play(self, from_generation, to_generation, last_model_elo=1200, start_train=False, writer=None):
        """coach.learn_to_play (coach.py:124-161)."""
        writer = writer or NullWriter()
        log = []
        g = from_generation
        while g <= to_generation:
            rec = {"generation": g}
            if not start_train:
                rec["selfplay"] = self.selfplay(g)
            start_train = False
            rec["last_batch_idx"] = self.train_nn(g, writer)
            if g > 0:
                elo_params = _get(self.params, "elo")
                if elo_params is not None:
                    _, last_model_elo, wins = sp.compute_elo(elo_params, [self.params, self.params], [max(0, g - 3), g],
                                                             (last_model_elo, last_model_elo), rows=self.rows, cols=self.cols,
                                                             device=self.device)
                    writer.add_scalar("elo", last_model_elo, rec["last_batch_idx"])
                    writer.add_scalar("wins", wins, rec["last_batch_idx"])
                    rec["elo"], rec["wins"] = last_model_elo, wins
            log.append(rec)
            g += 1
        return log

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None

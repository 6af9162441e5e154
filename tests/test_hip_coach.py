"""Generation loop with the replay resident in HBM (dotsboxesaz_amd/coach.py):
selfplay (HIP) -> ReplayStore -> device dataset/batches (HIP) -> optimizer step (torch on ROCm) ->
weights back into the HIP engine -> next generation's selfplay -> Elo match play."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Writer:
    def __init__(self):
        self.s = []

    def add_scalar(self, tag, v, i):
        self.s.append((tag, float(v), int(i)))

    def add_scalars(self, tag, d, i):
        for k, v in sorted(d.items()):
            self.s.append((tag + "/" + k, float(v), int(i)))


def test_three_generations_on_3x3(tmp_path):
    import torch
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd import train as T
    from dotsboxesaz_amd.coach import Coach
    params = dnn.resnet_params(3, 3, 32, 2, 4, 8)
    params["nn"]["model_class"] = dnn.ResNetZero
    params["nn"]["chkpts_filename"] = str(tmp_path / "model_gen{}.pt")
    params["nn"]["train_params"] = {"nb_epochs": 3, "train_batch_size": 128, "val_batch_size": 32, "lr": 1e-2,
                                    "lr_scheduler": T.GenerationLrScheduler({0: 1e-2, 2: 5e-3}),
                                    "optimizer_params": {"momentum": 0.9, "weight_decay": 1e-4},
                                    "pos_average": True, "train_split": 0.9, "max_samples_per_gen": 100000,
                                    "symmetries": None}
    params["self_play"] = {"num_games": 96, "reuse_mcts_tree": True, "noise": (0.8, 0.25),
                           "mcts": {"mcts_num_read": 24, "mcts_cpuct": (1.25, 19652), "temperature": {0: 1.0, 6: 0.02}}}
    params["elo"] = {"n_games": 16, "self_play_override": {"reuse_mcts_tree": False, "noise": (0.0, 0.0),
                                                           "mcts": {"mcts_num_read": 16}}}
    torch.manual_seed(0)
    np.random.seed(0)
    coach = Coach(params, 3, 3, n_slots=32)
    w = Writer()
    log = coach.learn_to_play(0, 2, writer=w)
    assert [r["generation"] for r in log] == [0, 1, 2]
    assert all(r["selfplay"]["rows"] > 96 * 8 for r in log)
    assert len(coach.store.chunks) == 3
    # generation 0 trains 0 epochs (min(2*0, nb_epochs)), generation 1 two, generation 2 three
    n_train = [len(c["train_locs"]) for c in coach.store.chunks]
    assert log[0]["last_batch_idx"] == 0 and log[1]["last_batch_idx"] > 0 and log[2]["last_batch_idx"] > log[1]["last_batch_idx"]
    cks = [torch.load(params["nn"]["chkpts_filename"].format(g), map_location="cpu", weights_only=True) for g in range(3)]
    assert [c["last_batch_idx"] for c in cks] == [r["last_batch_idx"] for r in log]
    k = "resnet.conv0.weight"
    assert not torch.equal(cks[0]["model_dict"][k], cks[1]["model_dict"][k])
    assert not torch.equal(cks[1]["model_dict"][k], cks[2]["model_dict"][k])
    losses = [v for t, v, _ in w.s if t == "loss/total/train"]
    assert len(losses) == log[2]["last_batch_idx"] and np.all(np.isfinite(losses))
    assert np.mean(losses[-3:]) < np.mean(losses[:3])            # it learns something on its own games
    assert [v for t, v, _ in w.s if t == "lr"] == [1e-2, 1e-2, 5e-3]
    assert "elo" in log[1] and "wins" in log[2] and sum(n_train) > 0
    # the engine now holds generation 2's weights: HIP predict == torch eval forward of the checkpoint
    model = dnn.ResNetZero(params)
    model.load_state_dict(cks[2]["model_dict"])
    model.train(False)
    x = np.random.RandomState(1).randint(0, 2, size=(64, 3, 4, 4)).astype(np.float32)
    x[:, 2] = 4.0
    with torch.no_grad():
        lp, v = T.training_forward(model, torch.tensor(x))
    p_hip, v_hip = coach.engine.predict(x)
    assert np.max(np.abs(p_hip - np.exp(lp.numpy()))) < 1e-4 and np.max(np.abs(v_hip - v.numpy())) < 1e-4
    coach.close()


def test_generation_loop_trains_through_the_hip_tower(tmp_path):
    """The same loop with a 64-channel ResNetZero: train.train() sends the residual blocks through csrc/train.hip
    (dbaz_trainer_*), the rest of the step through torch; the weights it produces drive the next generation's self-play."""
    import torch
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd import train as T
    from dotsboxesaz_amd import train_tower
    from dotsboxesaz_amd.coach import Coach
    params = dnn.resnet_params(3, 3, 64, 2, 4, 8)
    params["nn"]["model_class"] = dnn.ResNetZero
    params["nn"]["chkpts_filename"] = str(tmp_path / "model_gen{}.pt")
    params["nn"]["train_params"] = {"nb_epochs": 2, "train_batch_size": 128, "val_batch_size": 32, "lr": 1e-2,
                                    "lr_scheduler": T.GenerationLrScheduler({0: 1e-2}),
                                    "optimizer_params": {"momentum": 0.9, "weight_decay": 1e-4},
                                    "pos_average": True, "train_split": 0.9, "max_samples_per_gen": 100000, "symmetries": None}
    params["self_play"] = {"num_games": 96, "reuse_mcts_tree": True, "noise": (0.8, 0.25),
                           "mcts": {"mcts_num_read": 24, "mcts_cpuct": (1.25, 19652), "temperature": {0: 1.0, 6: 0.02}}}
    params["elo"] = {"n_games": 8, "self_play_override": {"reuse_mcts_tree": False, "noise": (0.0, 0.0), "mcts": {"mcts_num_read": 16}}}
    torch.manual_seed(0)
    np.random.seed(0)
    train_tower._trainers.clear()
    made = train_tower.handles_created
    coach = Coach(params, 3, 3, n_slots=32)
    w = Writer()
    log = coach.learn_to_play(0, 1, writer=w)
    assert [r["generation"] for r in log] == [0, 1] and log[1]["last_batch_idx"] > 0
    assert train_tower.handles_created == made + 1                # the HIP tower ran the training steps (one handle, one model)
    import gc
    gc.collect()
    assert len(train_tower._trainers) == 0                        # ... and went with the generation's model (train_nn builds a new one each time)
    losses = [v for t, v, _ in w.s if t == "loss/total/train"]
    assert len(losses) == log[1]["last_batch_idx"] and np.all(np.isfinite(losses)) and np.mean(losses[-3:]) < np.mean(losses[:3])
    ck = torch.load(params["nn"]["chkpts_filename"].format(1), map_location="cpu", weights_only=True)
    model = dnn.ResNetZero(params)
    model.load_state_dict(ck["model_dict"])
    assert int(ck["model_dict"]["resnet.resblocks.0.bn1.num_batches_tracked"]) == log[1]["last_batch_idx"]
    model.train(False)
    x = np.random.RandomState(1).randint(0, 2, size=(64, 3, 4, 4)).astype(np.float32)
    x[:, 2] = 4.0
    with torch.no_grad():
        lp, v = T.training_forward(model, torch.tensor(x))
    p_hip, v_hip = coach.engine.predict(x)                        # the engine holds the weights the HIP tower trained
    assert np.max(np.abs(p_hip - np.exp(lp.numpy()))) < 1e-4 and np.max(np.abs(v_hip - v.numpy())) < 1e-4
    coach.close()


def test_validation_on_the_hip_engine_equals_torchs_eval_forward(tmp_path):
    """NeuralNetWrapper.train runs its validation passes (nn.py:223-246) on the HIP inference engine by default.  For a network
    that has been trained for a generation, the per-batch validation losses (AlphaZeroLoss on log p, v) and the value accuracy
    count computed from the engine's (p, v) must equal those of torch's eval-mode forward of the same weights within 1e-4 --
    log(softmax) against log_softmax included -- and hip_validation=False must select torch's forward."""
    import torch
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd import train as T
    from dotsboxesaz_amd import train_data as TD
    from dotsboxesaz_amd.coach import Coach
    params = dnn.resnet_params(3, 3, 32, 2, 4, 8)
    params["nn"]["model_class"] = dnn.ResNetZero
    params["nn"]["chkpts_filename"] = str(tmp_path / "m_gen{}.pt")
    params["nn"]["train_params"] = {"nb_epochs": 2, "train_batch_size": 128, "val_batch_size": 32, "lr": 1e-2,
                                    "lr_scheduler": T.GenerationLrScheduler({0: 1e-2}), "optimizer_params": {"momentum": 0.9, "weight_decay": 1e-4},
                                    "pos_average": False, "train_split": 0.8, "max_samples_per_gen": 100000, "symmetries": None}
    params["self_play"] = {"num_games": 64, "reuse_mcts_tree": True, "noise": (0.8, 0.25),
                           "mcts": {"mcts_num_read": 16, "mcts_cpuct": (1.25, 19652), "temperature": {0: 1.0, 6: 0.02}}}
    params["elo"] = None
    torch.manual_seed(0)
    np.random.seed(0)
    coach = Coach(params, 3, 3, n_slots=32)
    w = Writer()
    coach.learn_to_play(0, 1, writer=w)                      # generation 1 trains two epochs, validating on the engine
    assert len([v for t, v, _ in w.s if t == "loss/total/eval"]) == 2
    ck = torch.load(params["nn"]["chkpts_filename"].format(1), map_location="cpu", weights_only=True)
    model = dnn.ResNetZero(params)
    model.load_state_dict(ck["model_dict"])
    model.train(False).cuda()
    coach.engine.load_state_dict(model.state_dict(), "resnet", **model.shape)
    val = coach.store.dataset(train=False, min_generation=0, pos_average=False)
    crit = T.AlphaZeroLoss()
    n = 0
    for boards, pi, z in T._batches(val, 32, False, None, torch.device("cuda", 0)):
        with torch.no_grad():
            lp_t, v_t = T.training_forward(model, boards)
        p, v = coach.engine.predict(boards.cpu().numpy())
        lp_h = torch.log(torch.from_numpy(p).clamp_min(1e-38)).cuda()
        v_h = torch.from_numpy(v).cuda()
        _, (lpi_t, lv_t) = crit(lp_t, v_t, pi, z)
        _, (lpi_h, lv_h) = crit(lp_h, v_h, pi, z)
        assert abs(lpi_t - lpi_h) < 1e-4 and abs(lv_t - lv_h) < 1e-4, (lpi_t, lpi_h, lv_t, lv_h)
        assert T._accuracy(v_t, z)[0] == T._accuracy(v_h, z)[0]
        n += 1
    assert n >= 2
    # the switch: hip_validation=False must not touch the engine's predict
    wrapper = dnn.NeuralNetWrapper(dnn.ResNetZero(params), params, engine=coach.engine)
    calls = []
    real = coach.engine.predict
    coach.engine.predict = lambda X: calls.append(1) or real(X)
    tr = coach.store.dataset(train=True, min_generation=0, pos_average=False)
    params["nn"]["train_params"]["symmetries"] = TD.SymmetriesGenerator(coach.engine)
    wrapper.train(tr, val, Writer(), 1, hip_validation=False)
    assert not calls
    wrapper.train(tr, val, Writer(), 1, hip_validation=True)
    assert calls
    coach.engine.predict = real
    coach.close()

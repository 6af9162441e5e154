#!/usr/bin/env python3
"""Wall clock of the generation loop on one GPU (coach.learn_to_play, coach.py:124-161), replay resident in HBM: self-play
(HIP) -> dataset + batches (HIP) -> optimizer steps (network, loss and SGD on HIP; torch holds the tensors) -> weights back
into the engine.  6x6, ResNetZero 20x64, 800 sims/move.  Prints one JSON line.

    python tools/generation_loop.py [games per generation] [last generation] [file to copy the last checkpoint to]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotsboxesaz_amd import nn as dnn  # noqa: E402
from dotsboxesaz_amd import train as T  # noqa: E402
from dotsboxesaz_amd.coach import Coach  # noqa: E402

games = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
last = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tmp = tempfile.mkdtemp()
params = dnn.resnet_params(6, 6, 64, 20)
params["nn"]["model_class"] = dnn.ResNetZero
params["nn"]["chkpts_filename"] = os.path.join(tmp, "model_gen{}.pt")
params["nn"]["train_params"] = {"nb_epochs": 10, "train_batch_size": 4096, "val_batch_size": 4096, "lr": 1e-2,
                                "lr_scheduler": T.GenerationLrScheduler({0: 1e-2}), "optimizer_params": {"momentum": 0.9, "weight_decay": 1e-4},
                                "pos_average": True, "train_split": 0.9, "max_samples_per_gen": 10 ** 9, "symmetries": None}
params["self_play"] = {"num_games": games, "reuse_mcts_tree": True, "noise": (0.8, 0.25),
                       "mcts": {"mcts_num_read": 800, "mcts_cpuct": (1.25, 19652), "temperature": {0: 1.0, 12: 0.02}}}
params["elo"] = None
torch.manual_seed(0)
np.random.seed(0)
coach = Coach(params, 6, 6, n_slots=min(games, 8192))
out = {"what": "generation loop on one MI355X: 6x6, ResNetZero 20x64, 800 sims/move, %d games per generation, window = all generations, "
               "min(2g, 10) epochs of 4096-sample batches" % games, "generations": []}
for g in range(last + 1):
    t0 = time.perf_counter()
    sp = coach.selfplay(g)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    last_idx0 = out["generations"][-1]["last_batch_idx"] if out["generations"] else 0
    last_idx = coach.train_nn(g, None)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    steps = last_idx - last_idx0
    out["generations"].append({"generation": g, "selfplay_s": t1 - t0, "games": games, "rows": sp.get("rows"), "games_per_sec": games / (t1 - t0),
                               "train_s": t2 - t1, "train_steps": steps, "ms_per_train_step_incl_data_and_validation": 1e3 * (t2 - t1) / max(1, steps),
                               "last_batch_idx": last_idx, "nn_evals": sp.get("nn_evals"), "f32_fallback_evals": sp.get("f32_fallback_evals"),
                               "f32_fallback_fraction": (sp.get("f32_fallback_evals") or 0) / max(1, sp.get("nn_evals") or 0),
                               "pool_resets": sp.get("pool_resets"), "moves_played": sp.get("moves_played")})
if len(sys.argv) > 3:  # keep the last generation's checkpoint (tests/test_hip_nn.py's trained-weights case reads it)
    import shutil
    shutil.copy(params["nn"]["chkpts_filename"].format(last), sys.argv[3])
coach.close()
print(json.dumps(out))  # (train() prints its epoch lines before this one: take the last line)

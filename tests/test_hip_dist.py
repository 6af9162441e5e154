"""RCCL path on one MI355X: the replay all-gather run with backend "nccl" in a
single-rank process group (two ranks cannot share a GPU under RCCL; world_size 2 is covered on
CPU with gloo in test_dist_gloo.py), and bench.py launched the way the driver launches it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from dotsboxesaz_amd.engine import Engine
from dotsboxesaz_amd.self_play import all_gather_rows, gather_replay, _DevBuf
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
e = Engine(3, 3, 8, mcts_num_read=20, evaluator="formula", seed=9)
e.selfplay_start(8, 0)
e.run()
ptr, n, rb = e.replay_rows_dev()
rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=torch.device("cuda", 0)).view(n, rb)
allrows, counts = all_gather_rows(rows, dist)
assert counts == [n], counts
assert allrows.is_cuda and allrows.shape == (n, rb)
assert torch.equal(allrows, rows)
total, ms = gather_replay(e, dist)
assert total == n
total2, _ = gather_replay(e, dist, synthetic_rows=n + 5)
assert total2 == n + 5
got = e.fetch_samples()
assert len(got["z"]) == n
e.close()
dist.destroy_process_group()
print("OK", n, rb)
"""


def _env(port):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def test_replay_all_gather_over_rccl_single_rank():
    r = subprocess.run([sys.executable, "-c", _CHILD % REPO], env=_env(29631), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-3000:]


def test_bench_under_torch_distributed_run():
    """The driver's launch line (one rank here): barrier + max-over-ranks timing + replay all-gather."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29632", os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "6",
           "--warmup", "2", "--board", "3", "--slots", "256", "--sims", "50", "--channels", "32", "--blocks", "2",
           "--no-cpu-baseline", "--no-f32-side-run"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["replay_allgather"]["rows"] >= 256 * 8  # the synthetic shard, or the finished rows if more
    assert np.isfinite(d["roofline"]["frac"])


def _bench(*flags):
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(flags), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_full_games_mode():
    d = _bench("--full-games", "48", "--slots", "16", "--board", "3", "--sims", "20", "--channels", "32", "--blocks", "2")
    assert d["metric"] == "selfplay_games_per_sec" and d["games"] == 48 and d["value"] > 0
    assert d["rows"] >= 48 * 8 and 0.0 <= d["cache_hit_fraction"] < 1.0


def test_bench_train_data_mode():
    d = _bench("--train-data", "50000", "--board", "6")
    assert d["metric"] == "train_data_rows_per_sec" and d["value"] > 0
    assert d["build_pos_average"]["rows_out"] < d["build_raw"]["rows_out"] == 50000
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "hbm"


def test_bench_train_step_mode():
    d = _bench("--train-step", "3", "--board", "3", "--blocks", "2")
    assert d["metric"] == "train_samples_per_sec" and d["value"] > 0 and d["torch_tower"]["ms_per_step"] > 0
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "mfma"


# ---- world_size 2 through the real engine: two processes share the one GPU of the box, the packed rows are
# exchanged with a host-staged gloo all-gather (RCCL cannot put two ranks on one device)
_CHILD2 = r"""
import os, sys, pickle
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
from dotsboxesaz_amd import nn as dnn
from dotsboxesaz_amd import self_play as sp
from dotsboxesaz_amd.coach import Coach
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", rank=rank, world_size=2)


class FormulaNN:
    kind = "formula"
    shape = {}
    def __init__(self, params): pass
    def load_parameters(self, g, to_device=None): pass


params = dnn.resnet_params(3, 3, 16, 1)
params["self_play"] = {"num_games": 11, "reuse_mcts_tree": True, "noise": [0.8, 0.25],
                       "mcts": {"mcts_num_read": 30, "mcts_cpuct": [1.25, 19652], "temperature": {0: 1.0, 6: 0.02}}}
df = sp.generate_games(None, 2, FormulaNN, 11, params, rows=3, cols=3, n_slots=4, dist=dist)
# Coach.selfplay, same exchange: the replay store of EVERY rank receives all games' rows
params["nn"]["model_class"] = dnn.ResNetZero
params["nn"]["train_params"] = {"train_split": 0.9}
torch.manual_seed(0)
coach = Coach(params, 3, 3, n_slots=4, dist=dist, nn_precision=0)
rec = coach.selfplay(0)
chunk = coach.store.chunks[0]["rows"]
games = sp.unpack_rows(chunk.cpu().numpy(), 48, 32)["game_idx"]
coach.close()
if rank == 0:
    with open(sys.argv[1], "wb") as f:
        pickle.dump((df, rec, np.unique(games).tolist(), len(games)), f)
dist.barrier()
dist.destroy_process_group()
print("OK")
"""


def test_generate_games_and_coach_selfplay_world2(tmp_path):
    """self_play.py:291-306 with two workers: the frame of rank 0 holds every game exactly once and is
    IDENTICAL to the single-process run (Philox streams are keyed by game index, not by rank)."""
    import pickle
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd import self_play as sp
    out = str(tmp_path / "rank0.pkl")
    procs = []
    for rank in range(2):
        env = _env(29641)
        env.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank))
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD2 % REPO, out], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        o, _ = p.communicate(timeout=900)
        assert p.returncode == 0 and "OK" in o, o[-3000:]
    with open(out, "rb") as f:
        df, rec, coach_games, coach_rows = pickle.load(f)  # written by this test's own child process

    class FormulaNN:
        kind = "formula"
        shape = {}

        def __init__(self, params):
            pass

        def load_parameters(self, g, to_device=None):
            pass

    params = dnn.resnet_params(3, 3, 16, 1)
    params["self_play"] = {"num_games": 11, "reuse_mcts_tree": True, "noise": [0.8, 0.25],
                           "mcts": {"mcts_num_read": 30, "mcts_cpuct": [1.25, 19652], "temperature": {0: 1.0, 6: 0.02}}}
    single = sp.generate_games(None, 2, FormulaNN, 11, params, rows=3, cols=3, n_slots=4)
    games = df.index.get_level_values("game_idx")
    assert sorted(set(games)) == list(range(11)) and not df.index.duplicated().any()
    assert df.equals(single)
    assert coach_games == list(range(11)) and rec["rows"] == coach_rows and rec["games"] == 11

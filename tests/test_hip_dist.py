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
    assert d["value"] > 0 and d["replay_allgather"]["rows"] == 256 * 8
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

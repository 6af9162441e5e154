#!/usr/bin/env python3
"""Headline benchmark: MCTS node-expansions/s (+ self-play games/s) on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch: every concurrent game performs one
simulation (select leaf -> policy/value network on the coalesced leaves -> expand + backup,
plus the driver: move sampling, re-rooting with tree reuse, game retirement/refill).
Workload at N=1 = BASELINE.json configs[2]: 6x6 board, 8192 concurrent games, 800 sims/move,
ResNetZero 20x64 (random init), Dirichlet noise (0.8, 0.25), tree reuse; games start from a
synthetic mid-game population (uniformly random legal plies) so that all game phases are
present.  For N>1 every rank runs the same per-GPU workload on its own game-index range
(weak scaling; games are independent, no data-path collective) and the ranks all-gather
their replay rows over RCCL at the end.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/f16


def cpu_baseline(rows, cols, sims, budget_s, channels, blocks):
    """Oracle (C tree/rules restatement + torch fp32 CPU network, batch 1) timed on the host
    cores: the reference's CPU path, bounded sample."""
    import torch
    from oracle import oracle as O
    from oracle import nn_ref
    torch.manual_seed(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))  # batch-1 convs do not scale past a few cores
    model = nn_ref.ResNetZeroRef(rows, cols, channels, blocks)
    model.train(False)
    d = O.dims(rows, cols)

    def fn(dd, s):
        x = O.features(dd, s)[None].astype(np.float32)
        p, v = nn_ref.predict_sync(model, x)
        return p[0], v[0]

    ev = O.Evaluator(fn)
    tree = O.Tree(d, O.new_state(d))
    rng = np.random.RandomState(0)
    t0 = time.perf_counter()
    n0 = 0
    done = 0
    while time.perf_counter() - t0 < budget_s:
        noise = rng.dirichlet(np.full(d.A, 0.8))
        left = sims
        while left > 0 and time.perf_counter() - t0 < budget_s:
            k = min(40, left)
            vis = tree.search(k, ev, dirichlet=(0.8, 0.25) if left == sims else (0.0, 0.0), noise=noise)
            left -= k
        done = tree.counters()[0]
        if left == 0:
            tree.advance(int(np.argmax(vis)), True)
            if tree.is_terminal:
                n0 += done
                tree = O.Tree(d, O.new_state(d))
    dt = time.perf_counter() - t0
    total = n0 + tree.counters()[0]
    return dict(value=total / dt, unit="expansions/s", cores=int(torch.get_num_threads()), kind="port",
                sample="%dx%d, 1 game, %d sims/move, sequential search, torch fp32 CPU ResNetZero %dx%d batch 1, "
                       "%.0f s (%d expansions)" % (rows, cols, sims, blocks, channels, dt, total))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--board", type=int, default=6)
    ap.add_argument("--slots", type=int, default=8192)
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=20)
    ap.add_argument("--evaluator", default="resnet", choices=["resnet", "formula", "uniform"])
    ap.add_argument("--precision", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd import nn as dnn

    rows = cols = args.board
    eng = Engine(rows, cols, args.slots, mcts_num_read=args.sims, noise=(0.8, 0.25), reuse_tree=True,
                 evaluator=args.evaluator, seed=1000 + rank, device=local_rank, nn_precision=args.precision)
    if args.evaluator == "resnet":
        torch.manual_seed(0)
        model = dnn.ResNetZero(dnn.resnet_params(rows, cols, args.channels, args.blocks))
        eng.load_state_dict(model.state_dict(), "resnet", **model.shape)
    # synthetic mid-game population: slot i starts (i*37 mod 0.7E) random legal plies into a game
    span = max(1, int(0.7 * eng.E))
    eng.selfplay_fastforward((np.arange(args.slots) * 37) % span)
    eng.selfplay_start(1 << 40, rank * (1 << 32))

    def sync_all():
        eng.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    eng.step(args.warmup)
    sync_all()
    c0 = eng.counters()
    eng.timing_begin()
    t0 = time.perf_counter()
    eng.step(args.steps)
    eng.timing_end()
    sync_all()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    if c1["error_slots"]:
        raise SystemExit("engine reported %d slots in error (node pool exhausted?)" % c1["error_slots"])
    exp = c1["expansions"] - c0["expansions"]
    evals = c1["nn_evals"] - c0["nn_evals"]
    spath = c1["sum_path"] - c0["sum_path"]
    term = c1["terminal_leaves"] - c0["terminal_leaves"]
    tot = torch.tensor([float(exp), float(evals), dt], dtype=torch.float64)
    if dist is not None:
        tdev = tot.cuda()
        mx = tdev.clone()
        dist.all_reduce(tdev, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        exp_all, evals_all, dt_max = float(tdev[0]), float(tdev[1]), float(mx[2])
    else:
        exp_all, evals_all, dt_max = float(exp), float(evals), dt

    # replay all-gather at iteration end (multi-GPU): whatever finished + a fixed synthetic shard
    gather = None
    if dist is not None:
        from dotsboxesaz_amd.self_play import gather_replay
        rows_total, ms = gather_replay(eng, dist, synthetic_rows=args.slots * 8)
        gather = dict(rows=rows_total, ms=ms)

    if rank == 0:
        HW = (rows + 1) * (cols + 1)
        conv_flops = 2.0 * HW * 9 * args.channels * args.channels  # per sample per tower conv launch
        launches = args.steps  # one fused-tower launch per step (all 2*blocks conv layers)
        out = {
            "metric": "mcts_node_expansions_per_sec", "value": exp_all / dt_max, "unit": "expansions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == 0 else "f32 (f16x3 split MFMA, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%dx%d board, %d concurrent games/GPU, %d sims/move, evaluator=%s %dx%d "
                                   "random-init, noise (0.8,0.25), tree reuse, mid-game start population"
                                   % (rows, cols, args.slots, args.sims, args.evaluator, args.blocks, args.channels),
                       "baseline_config": "configs[2]" if (args.board, args.slots, args.sims) == (6, 8192, 800) else "custom",
                       "parallelism": "games sharded, %d rank(s)" % world},
            "per_gpu": exp_all / dt_max / world,
            "nn_evals_per_sec": evals_all / dt_max,
            "mean_path_len": spath / max(1, exp), "terminal_leaf_fraction": term / max(1, exp),
            "pool_high_water": c1["pool_high_water"], "nodes_per_slot": eng.cfg.nodes_per_slot or 6 * (args.sims + 2),
            "moves_played": c1["moves_played"] - c0["moves_played"],
        }
        # games/s: expansions/s divided by the measured mean expansions of a full game
        # (DESIGN.md "Measurement"; 6x6 @ 800 sims: 63.2k)
        exp_per_game = {(6, 800): 63201.0, (3, 100): 1901.0, (9, 1600): 280481.0}.get((args.board, args.sims))
        if exp_per_game:
            out["games_per_sec_est"] = out["value"] / exp_per_game
            out["expansions_per_game_assumed"] = exp_per_game
        if args.evaluator == "resnet" and c1["ms_nn_tower"] > 0:
            ach = evals * conv_flops * 2 * args.blocks / (c1["ms_nn_tower"] * 1e-3) / 1e12
            peak = F32_MFMA_PEAK_TFLOPS if args.precision == 0 else F16_MFMA_PEAK_TFLOPS / 3.0
            out["roofline"] = {"bound": "mfma", "kernel": "k_tower_%s<%d> (2*%d conv3x3 layers fused, LDS-resident)" % ("f32" if args.precision == 0 else "f16x3", args.channels, args.blocks), "achieved": ach, "peak": peak,
                               "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                               "avg_launch_us": 1e3 * c1["ms_nn_tower"] / launches,
                               "flops_per_launch": evals / args.steps * conv_flops * 2 * args.blocks,
                               "tower_ms_per_step": c1["ms_nn_tower"] / args.steps,
                               "tree_and_heads_ms_per_step": (c1["ms_total"] - c1["ms_nn_tower"]) / args.steps}
        else:
            # tree kernels only: HBM roofline with SURVEY 8d's algorithmic bytes per simulation
            A = 2 * HW
            L = spath / max(1, exp)
            bps = (L - 1) * (12 * A + A / 4 + 12) + L * 16 + (12 * A + (A + 7) // 8 + 8) + 2 * (3 * HW * 4) + 4 * A + 4
            ach = exp * bps / (c1["ms_total"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_select+k_expand_backup", "achieved": ach, "peak": 8000.0,
                               "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "bytes_per_sim": bps}
        if gather:
            out["replay_allgather"] = gather
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rows, cols, args.sims, args.cpu_seconds, args.channels, args.blocks)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()

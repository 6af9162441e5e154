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


def host_cores():
    """Cores this process may use: the affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_worker(job):
    """One self-play game loop of the oracle on ONE core (torch intra-op threads = 1): C tree / rules
    restatement + torch fp32 CPU ResNetZero at batch 1 with the reference's (p, v) cache by position.
    Returns (expansions, seconds)."""
    rows, cols, sims, budget_s, channels, blocks, seed = job
    import torch
    torch.set_num_threads(1)
    from oracle import oracle as O
    from oracle import nn_ref
    torch.manual_seed(0)
    model = nn_ref.ResNetZeroRef(rows, cols, channels, blocks)
    model.train(False)
    d = O.dims(rows, cols)
    cache = {}  # utils/proxies.py:35-43

    def fn(dd, s):
        f = O.features(dd, s)
        k = f.tobytes()
        hit = cache.get(k)
        if hit is None:
            p, v = nn_ref.predict_sync(model, f[None].astype(np.float32))
            hit = cache[k] = (p[0], v[0])
        return hit

    ev = O.Evaluator(fn)
    tree = O.Tree(d, O.new_state(d))
    rng = np.random.RandomState(seed)
    t0 = time.perf_counter()
    n0 = 0
    while time.perf_counter() - t0 < budget_s:
        noise = rng.dirichlet(np.full(d.A, 0.8))
        left = sims
        vis = None
        while left > 0 and time.perf_counter() - t0 < budget_s:
            k = min(40, left)
            vis = tree.search(k, ev, dirichlet=(0.8, 0.25) if left == sims else (0.0, 0.0), noise=noise)
            left -= k
        if left == 0:
            # temperature 1 for the first moves like the reference's schedule; argmax afterwards
            tree.advance(int(np.argmax(vis)), True)
            if tree.is_terminal:
                n0 += tree.counters()[0]
                tree = O.Tree(d, O.new_state(d))
    return n0 + tree.counters()[0], time.perf_counter() - t0


def cpu_baseline(rows, cols, sims, budget_s, channels, blocks, max_procs=64):
    """The reference's CPU path as the oracle restates it, one process per host core (the reference
    runs mp.cpu_count()-1 worker processes, self_play.py:292), bounded sample: every process plays its
    own game for budget_s seconds.  Must run BEFORE this process touches the GPU (it forks workers)."""
    import multiprocessing as mp
    cores = host_cores()
    procs = min(cores, max_procs)
    jobs = [(rows, cols, sims, budget_s, channels, blocks, i) for i in range(procs)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    total = sum(r[0] for r in res)
    rate = sum(r[0] / r[1] for r in res)
    return dict(value=rate, unit="expansions/s", cores=procs, host_cores=cores, kind="port", per_core=rate / procs,
                sample="%dx%d, %d sims/move, %d processes x 1 thread (one per host core), each its own game: sequential search "
                       "(C oracle) + torch fp32 CPU ResNetZero %dx%d at batch 1 with the reference's (p, v) cache by position, "
                       "%.0f s each (%d expansions in all, %.0f s wall incl. process start)"
                       % (rows, cols, sims, procs, blocks, channels, budget_s, total, wall))


PMC_TOWER_PROFILE = os.path.join(REPO, "profiles", "r03_pmc_tower_driver_window.json")


def tower_traffic(profile_path, lib_nn_hash, evals_per_launch):
    """roofline.traffic of the main k_tower launch from the committed PMC summary (tools/pmc_driver_window.sh; PMC passes cannot run
    inside this process) -- but only when that summary was measured on the build of the network kernels that is loaded now: the
    summary stores dbaz_build_info()'s nn= hash of the library it ran on.  Returns (traffic or None, note, mfma_busy_frac or None)."""
    if not os.path.exists(profile_path):
        return None, "no PMC summary at %s" % os.path.relpath(profile_path, REPO), None
    t = json.load(open(profile_path))
    have = (t.get("build") or {}).get("nn")
    if not have or have != lib_nn_hash:
        return None, ("traffic withheld: %s was measured on network kernels nn=%s, the loaded library is nn=%s -- rerun "
                      "tools/pmc_driver_window.sh on this build" % (os.path.relpath(profile_path, REPO), have, lib_nn_hash)), None
    per_eval = t["traffic_bytes_per_launch"] / t["evals_per_launch"]
    note = ("(2*FETCH_SIZE + WRITE_SIZE)*1024 B of the main k_tower launch, averaged over the timed launches of `bench.py --steps %s "
            "--warmup %s` under rocprofv3 --pmc (%s, same build nn=%s: %.1f MB at %.0f evaluations per launch), scaled to this run's "
            "%.0f evaluations per launch; algorithmic bytes per launch = features in + head activations out + weights once = %.1f MB"
            % (t.get("steps", "?"), t.get("warmup", "?"), os.path.relpath(profile_path, REPO), have, t["traffic_bytes_per_launch"] / 1e6,
               t["evals_per_launch"], evals_per_launch, (evals_per_launch * (588 + 6272) + 5.9e6) / 1e6))
    return per_eval * evals_per_launch, note, t.get("mfma_busy_frac")


def make_engine(args, precision, rank, local_rank, torch, slots=None):
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd import nn as dnn
    rows = cols = args.board
    eng = Engine(rows, cols, slots or args.slots, mcts_num_read=args.sims, noise=(0.8, 0.25), reuse_tree=True,
                 evaluator=args.evaluator, seed=1000 + rank, device=local_rank, nn_precision=precision,
                 nodes_per_slot=args.nodes_per_slot, transposition_cache=not args.no_tt, debug_flags=args.debug_flags)
    if args.evaluator == "resnet":
        torch.manual_seed(0)
        model = dnn.ResNetZero(dnn.resnet_params(rows, cols, args.channels, args.blocks))
        sd = model.state_dict()
        if args.zero_weights:  # DVFS diagnosis only (MI355X_MICROARCH.md, give-back item 1): all-zero MFMA operands
            sd = {k: (v * 0 if ("conv" in k or "fc" in k) else v) for k, v in sd.items()}
        eng.load_state_dict(sd, "resnet", **model.shape)
    elif args.evaluator == "simplenn":
        torch.manual_seed(0)
        eng.load_state_dict(dnn.SimpleNN().state_dict(), "simplenn")
    return eng


def run_engine(args, precision, steps, warmup, rank, local_rank, world, dist, torch, population=None):
    """One timed region on this rank's GPU; returns the raw measurements.

    Population (synthetic input, built before the W warm-up steps; --population):
      games   (default) slot i plays the first (i*37 mod game length) plies of a game with --quick-reads simulations per
              move (real search, network, temperature, tree reuse -- only a smaller budget), then its first full search is
              cut to (i*61 mod sims)+1 reads; after the untimed preparation steps every slot is somewhere inside a full
              search of a mid-game move with a re-rooted, reused tree: the steady state of a long self-play run with
              slot refill, on positions of the kind search-based play reaches;
      random  the same with uniformly random legal plies instead of quick play (cheaper to build, but random play leaves
              positions with more terminal leaves and transpositions than games have: 12.7 % / 38.6 % against the 5.3 % /
              30.6 % measured over complete games);
      fresh   round 1: random plies, empty trees (path length 2.9, no terminal leaves in a 20-step window)."""
    eng = make_engine(args, precision, rank, local_rank, torch)
    prep = 0
    population = population or args.population
    if population == "games":
        # slot i: the first (i*37 mod E_mean) plies of its game at args.quick_reads reads per move, then a first full search
        # cut to (i*61 mod sims)+1 reads; E_mean = the mean length of a game (plies), so the slots sample a game's plies uniformly
        span = max(1, int(args.game_plies or 0.97 * eng.E))
        plies = (np.arange(args.slots) * 37) % span
        eng.selfplay_quickplay(plies, args.quick_reads)
        eng.selfplay_stagger((np.arange(args.slots) * 61) % max(1, args.sims) + 1)
        prep = int(plies.max()) * (args.quick_reads + 2) + args.sims + 64
    elif population == "random":
        span = max(1, int(0.7 * eng.E))
        eng.selfplay_fastforward((np.arange(args.slots) * 37) % span)
        eng.selfplay_stagger((np.arange(args.slots) * 61) % max(1, args.sims) + 1)
        prep = args.sims + 64
    else:  # "fresh": round 1's population, every tree empty
        eng.selfplay_fastforward((np.arange(args.slots) * 37) % max(1, int(0.7 * eng.E)))
    eng.selfplay_start(1 << 40, rank * (1 << 32))
    eng.step(prep)

    def sync_all():
        eng.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    eng.step(warmup)
    c0 = eng.counters()  # (before the bracket: the GPU should idle as briefly as possible ahead of the timed steps -- after an
    sync_all()           #  idle gap the chip needs ~25 launches to return to its sustained clock, profiles/r02_post_idle_ramp.txt)
    eng.timing_begin()
    t0 = time.perf_counter()
    eng.step(steps)
    eng.timing_end()
    sync_all()
    dt = time.perf_counter() - t0
    c1 = eng.counters()
    if c1["error_slots"]:
        raise SystemExit("engine reported %d slots in error (node pool exhausted?)" % c1["error_slots"])
    m = {k: c1[k] - c0[k] for k in ("expansions", "nn_evals", "sum_path", "terminal_leaves", "moves_played", "cache_hits",
                                    "f32_fallback_evals", "pool_resets")}
    m.update(dt=dt, ms_total=c1["ms_total"], ms_nn_tower=c1["ms_nn_tower"], pool_high_water=c1["pool_high_water"],
             nodes_per_slot=eng.nodes_per_slot, E=eng.E, prep_steps=prep)
    if dist is not None:
        tdev = torch.tensor([float(m["expansions"]), float(m["nn_evals"]), dt], dtype=torch.float64).cuda()
        mx = tdev.clone()
        dist.all_reduce(tdev, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        m.update(exp_all=float(tdev[0]), evals_all=float(tdev[1]), dt_max=float(mx[2]))
    else:
        m.update(exp_all=float(m["expansions"]), evals_all=float(m["nn_evals"]), dt_max=dt)
    return eng, m


def play_complete_games(args, n_games, slots, rank, local_rank, torch, budget_s=0.0):
    """Direct games/s: n_games complete games from the empty board (slots refilled as games end), samples
    fetched, wall clock around everything after engine construction.  budget_s > 0 stops a run that takes
    longer (reported as partial: the rate then covers the games finished so far)."""
    eng = make_engine(args, args.precision, rank, local_rank, torch, slots=slots)
    eng.sync()
    t0 = time.perf_counter()
    eng.selfplay_start(n_games, rank * n_games)
    rows_seen = 0
    partial = False
    while True:
        eng._ck(eng._L.dbaz_run(eng.h, 1024))
        c = eng.counters()
        if c["active_slots"] == 0:
            break
        if c["blocked_slots"] > 0 or c["rows_ready"] > (1 << 20):
            rows_seen += len(eng._fetch_once()["z"])
        if budget_s > 0 and time.perf_counter() - t0 > budget_s:
            partial = True
            break
    rows_seen += len(eng._fetch_once()["z"])
    dt = time.perf_counter() - t0
    c = eng.counters()
    out = {"value": c["games_finished"] / dt, "unit": "games/s", "seconds": dt, "games": c["games_finished"],
           "games_requested": n_games, "slots": slots, "partial": partial,
           "expansions": c["expansions"], "expansions_per_sec": c["expansions"] / dt,
           "expansions_per_game": c["expansions"] / max(1, c["games_finished"]),
           "rows": rows_seen, "rows_per_game": rows_seen / max(1, c["games_finished"]),
           "mean_path_len": c["sum_path"] / max(1, c["expansions"]),
           "terminal_leaf_fraction": c["terminal_leaves"] / max(1, c["expansions"]),
           "cache_hit_fraction": c["cache_hits"] / max(1, c["expansions"]),
           "pool_high_water": c["pool_high_water"], "pool_resets": c["pool_resets"], "steps": c["steps"],
           "nn_evals": c["nn_evals"], "f32_fallback_evals": c["f32_fallback_evals"],
           "f32_fallback_fraction": c["f32_fallback_evals"] / max(1, c["nn_evals"])}
    eng.close()
    return out


def full_games(args, rank, local_rank, world, torch):
    rows = cols = args.board
    out = {"metric": "selfplay_games_per_sec", "n_gpus": 1, "higher_is_better": True, "data": "synthetic"}
    out.update(play_complete_games(args, args.full_games, args.slots, rank, local_rank, torch))
    out["config"] = {"workload": "%dx%d board, %d complete games on %d slots, %d sims/move, evaluator=%s %dx%d, precision %d"
                                 % (rows, cols, args.full_games, args.slots, args.sims, args.evaluator, args.blocks,
                                    args.channels, args.precision)}
    print(json.dumps(out), flush=True)


def single_tree_bench(args, local_rank, torch):
    """SURVEY 8f-4 (players.AZPlayer): ONE tree searched for args.single_tree seconds with args.pending simulations in
    flight (virtual loss, one batched network launch per wave).  Reads per second on the empty board."""
    eng_args = argparse.Namespace(**vars(args))
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd import nn as dnn
    rows = cols = args.board
    res = {}
    for K in sorted({1, 8, args.pending}):
        eng = Engine(rows, cols, 1, mcts_num_read=args.sims, evaluator=args.evaluator, device=local_rank, nn_precision=args.precision,
                     nodes_per_slot=1 << 20, max_pending_evals=max(K, 2))
        if args.evaluator == "resnet":
            torch.manual_seed(0)
            model = dnn.ResNetZero(dnn.resnet_params(rows, cols, args.channels, args.blocks))
            eng.load_state_dict(model.state_dict(), "resnet", **model.shape)
        eng.set_pending(K)
        eng.set_positions(None)
        t0 = time.perf_counter()
        eng.search_timed(args.single_tree, num_reads=2 ** 31 - 1)
        dt = time.perf_counter() - t0
        r, c = eng.roots(), eng.counters()
        reads = int(r["root_nv"][0]) - 1
        res[str(K)] = {"reads": reads, "seconds": dt, "reads_per_sec": reads / dt, "nn_evals": c["nn_evals"],
                       "shared_pending_leaves": c["cache_hits"], "mean_path_len": c["sum_path"] / max(1, c["expansions"]),
                       "most_visited_share": float(r["visits"][0].max()) / max(1, reads), "pool_high_water": c["pool_high_water"], "pool_resets": c["pool_resets"]}
        eng.close()
    out = {"metric": "single_tree_reads_per_sec", "value": res[str(args.pending)]["reads_per_sec"], "unit": "reads/s", "n_gpus": 1,
           "higher_is_better": True, "data": "synthetic", "by_pending": res,
           "config": {"workload": "%dx%d, ONE tree from the empty board, %.1f s wall clock, evaluator=%s %dx%d precision %d, "
                                  "max_pending_evals=%d" % (rows, cols, args.single_tree, args.evaluator, args.blocks, args.channels,
                                                            args.precision, args.pending)}}
    print(json.dumps(out), flush=True)


def train_step_bench(args, local_rank):
    """NeuralNetWrapper.train's step (nn.py:203-221) at the reference's batch size (configuration.py:61), the whole
    network, the loss and the SGD update on the HIP kernels of csrc/train.hip against the same step with the network on torch
    (MIOpen / rocBLAS)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import train_step_time as TS
    batch = 4096
    hip = TS.measure(batch, True, args.train_step, args.board, args.channels, args.blocks, local_rank)
    ref = TS.measure(batch, False, max(3, args.train_step // 2), args.board, args.channels, args.blocks, local_rank)
    HW = (args.board + 1) ** 2
    # conv3x3 forward + input gradient + weight gradient of the 2*blocks tower layers (2*MAC each), counted once
    flops = 3.0 * 2 * args.blocks * 2.0 * batch * HW * args.channels * args.channels * 9
    tf = flops / (hip["ms_per_step"] * 1e-3) / 1e12
    out = {"metric": "train_samples_per_sec", "value": hip["samples_per_sec"], "unit": "samples/s", "n_gpus": 1, "steps": args.train_step,
           "ms_per_step": hip["ms_per_step"], "higher_is_better": True, "vs_baseline": None,
           "dtype": "f32 via f16x3 (hi,lo)-split MFMA (tower convs), exact f32 (stem, heads, loss, SGD), f64 batch statistics",
           "data": "synthetic", "config": {"workload": "%dx%d, ResNetZero %dx%d, batch %d, SGD momentum 0.9 wd 1e-4, batches from k_make_batch"
                                           % (args.board, args.board, args.blocks, args.channels, batch)},
           "torch_tower": {"ms_per_step": ref["ms_per_step"], "samples_per_sec": ref["samples_per_sec"]},
           "speedup_vs_torch_tower": ref["ms_per_step"] / hip["ms_per_step"],
           "roofline": {"bound": "mfma", "kernel": "k_conv_t + k_wgrad_h3 (tower conv forward, input gradient, weight gradient), whole step in the time",
                        "achieved": tf, "peak": 2500.0 / 3, "unit": "TFLOP/s", "frac": tf / (2500.0 / 3), "flops_per_step": flops,
                        "traffic": None,
                        "note": "the step moves ~30 GB of activations through HBM (4 ms at peak): the elementwise passes and the convs' "
                                "un-overlapped load/store phases, not the MFMA pipe, set the time (EXPERIMENTS.md 3)"}}
    print(json.dumps(out), flush=True)


def train_data_bench(args, local_rank, torch):
    """Training DATA path on synthetic replay rows resident in HBM: dataset build
    (stage + stable radix sort + Kahan group means = HDFStoreDataset with pos_average) and batch
    assembly (gather + symmetry).  HBM roofline; algorithmic bytes per row stated in DESIGN.md."""
    from dotsboxesaz_amd.engine import Engine
    rows = cols = args.board
    e = Engine(rows, cols, 4, mcts_num_read=8, evaluator="formula", nodes_per_slot=64, device=local_rank)
    dev = torch.device("cuda", local_rank)
    n, F, A, HW, rb = args.train_data, e.F, e.A, e.H * e.W, e.row_bytes
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    # positions drawn from a pool a quarter the size of the row count (duplicates as in real replay windows)
    pool = max(1, n // 4)
    edges = (torch.rand((pool, A), device=dev, generator=g) < 0.4).to(torch.int16)
    sent = torch.zeros(A, dtype=torch.bool, device=dev)
    sent.view(2, e.H, e.W)[1, e.H - 1, :] = True
    sent.view(2, e.H, e.W)[0, :, e.W - 1] = True
    edges[:, sent] = 0
    plane2 = torch.randint(0, 2 * rows * cols, (pool, 1), device=dev, generator=g, dtype=torch.int16).expand(pool, HW)
    xpool = torch.cat([edges, plane2], dim=1).contiguous()
    pick = torch.randint(0, pool, (n,), device=dev, generator=g)
    x = xpool[pick]
    vis = torch.randint(0, 50, (n, A), device=dev, generator=g, dtype=torch.int32) * (1 - x[:, :A].to(torch.int32))
    vis[:, 0] += 1
    buf = torch.zeros((n, rb), dtype=torch.uint8, device=dev)
    buf[:, 25] = torch.randint(0, 3, (n,), device=dev, generator=g).to(torch.uint8) - 1  # z in {-1,0,1} as int8 bits
    buf[:, 28:28 + 2 * F] = x.view(torch.uint8).view(n, 2 * F)
    buf[:, 28 + 2 * F:28 + 2 * F + 4 * A] = vis.view(torch.uint8).view(n, 4 * A)
    del x, vis, xpool, edges, pick
    torch.cuda.synchronize()
    out = {"metric": "train_data_rows_per_sec", "unit": "rows/s", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
           "config": {"workload": "%dx%d, %d packed replay rows in HBM (%d B/row), position pool %d" % (rows, cols, n, rb, pool)}}
    for avg in (False, True):
        e.dataset_begin()
        e.dataset_add_rows(buf)
        e.dataset_finish(avg)  # warm-up (allocations)
        t0 = time.perf_counter()
        e.dataset_begin()
        e.dataset_add_rows(buf)
        m = e.dataset_finish(avg)
        dt = time.perf_counter() - t0
        # algorithmic bytes: read the packed row, write x + pi + z of the dataset row
        byts = n * rb + m * (2 * F + 4 * A + 4)
        out["build_pos_average" if avg else "build_raw"] = {"rows_in": n, "rows_out": m, "seconds": dt, "rows_per_sec": n / dt,
                                                             "GBps_algorithmic": byts / dt / 1e9}
    m = out["build_pos_average"]["rows_out"]
    rs = np.random.RandomState(0)
    res = {}
    for B in (4096, 262144):
        idx = rs.randint(0, m, size=B).astype(np.int32)
        e.dataset_batch(idx, 5)
        reps = 50 if B <= 4096 else 10
        t0 = time.perf_counter()
        for r in range(reps):
            e.dataset_batch(idx, r & 7)
        dt = (time.perf_counter() - t0) / reps
        byts = B * ((2 * F + 4 * A + 4) + (4 * F + 4 * A + 4))  # read dataset row, write float32 boards + pi + z
        res[str(B)] = {"seconds_per_batch": dt, "rows_per_sec": B / dt, "GBps_algorithmic": byts / dt / 1e9}
    out["batch"] = res
    big = res["262144"]
    out["value"] = big["rows_per_sec"]
    traffic = None
    pmc = os.path.join(REPO, "profiles", "r01_pmc_train_data.json")
    if os.path.exists(pmc) and (args.board, n) == (6, 2000000):
        traffic = json.load(open(pmc))["kernels"].get("k_make_batch", {}).get("traffic_bytes_max_launch")
    out["roofline"] = {"bound": "hbm", "kernel": "k_make_batch (gather + symmetry LUT, one wave per row)",
                       "achieved": big["GBps_algorithmic"], "peak": 8000.0, "unit": "GB/s", "frac": big["GBps_algorithmic"] / 8000.0,
                       "traffic": traffic, "algorithmic_bytes_per_launch": 262144 * ((2 * F + 4 * A + 4) + (4 * F + 4 * A + 4)), "note": "wall clock per call incl. index upload, allocation of the output tensors and the "
                                                "stream sync that orders the batch before torch"}
    print(json.dumps(out), flush=True)
    e.close()


def tower_roofline(args, m, steps, precision):
    HW = (args.board + 1) ** 2
    conv_flops = 2.0 * HW * 9 * args.channels * args.channels  # per sample per conv3x3 layer
    ach = m["nn_evals"] * conv_flops * 2 * args.blocks / (m["ms_nn_tower"] * 1e-3) / 1e12
    peak = F32_MFMA_PEAK_TFLOPS if precision == 0 else F16_MFMA_PEAK_TFLOPS / 3.0
    return {"bound": "mfma",
            "kernel": "k_tower<%d,*,%s> (conv0 + 2*%d conv3x3 + head 1x1 convs fused, LDS-resident)"
                      % (args.channels, {0: "f32", 1: "f16x3 on 16x16x32 (two cout tiles per wave at 64 channels)", 2: "f16x3 on 32x32x16", 3: "f16x3 on 16x16x32, two cout tiles per wave", 4: "f16x3 on 16x16x32, one cout tile per wave"}.get(precision, "f16x3, A/B variant %d of the debug build" % precision), args.blocks),
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            "peak_note": ("dense f32 MFMA" if precision == 0 else
                          "dense f16 MFMA / 3 (three f16 MFMAs per f32-grade product); algorithmic flops counted once"),
            "avg_launch_us": 1e3 * m["ms_nn_tower"] / steps,
            "flops_per_launch": m["nn_evals"] / steps * conv_flops * 2 * args.blocks,
            "tower_ms_per_step": m["ms_nn_tower"] / steps,
            "tree_and_heads_ms_per_step": (m["ms_total"] - m["ms_nn_tower"]) / steps}


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
    ap.add_argument("--evaluator", default="resnet", choices=["resnet", "simplenn", "formula", "uniform"])
    ap.add_argument("--precision", type=int, default=1, choices=list(range(15)),
                    help="0 = exact f32 MFMA; 1 = f16x3 error-compensated MFMA (f32-grade, default); 2 / 3 / 4 = A/B tilings of the "
                         "f16x3 layer, debug build only (DBAZ_LIB=dotsboxesaz_amd/libdbaz_hip_debug.so; EXPERIMENTS.md)")
    ap.add_argument("--debug-flags", type=int, default=0, help="dbaz_config.debug_flags (1 = early join, 2 = no f32 fallback launch)")
    ap.add_argument("--nodes-per-slot", type=int, default=0, help="tree node pool per game (0 = engine default 10*(sims+2))")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-side-run", action="store_true")
    ap.add_argument("--zero-weights", action="store_true", help="diagnosis: all conv/fc weights zero (clock under load vs operand data)")
    ap.add_argument("--no-tt", action="store_true", help="switch the per-game transposition table off (kernel measurements)")
    ap.add_argument("--population", default="games", choices=["games", "random", "fresh"],
                    help="how the slots' starting positions and trees are prepared before the warm-up (see run_engine)")
    ap.add_argument("--fresh-population", action="store_true", help="same as --population fresh")
    ap.add_argument("--quick-reads", type=int, default=48, help="--population games: simulations per move of the opening plies")
    ap.add_argument("--game-plies", type=int, default=0, help="--population games: plies sampled (0 = 0.97 E, the mean game length)")
    ap.add_argument("--games-leg", type=int, default=-1,
                    help="complete games played (per rank) AFTER the timed steps for the directly measured games/s of the "
                         "metric; -1 = one game per slot (the BASELINE config's game count), 0 = skip")
    ap.add_argument("--games-leg-budget", type=float, default=240.0, help="wall-clock cap of the games leg in seconds")
    ap.add_argument("--full-games", type=int, default=0,
                    help="instead of timing K steps, play this many COMPLETE games from the empty board and report "
                         "games/s and expansions/game measured directly (one JSON line, metric selfplay_games_per_sec)")
    ap.add_argument("--single-tree", type=float, default=0.0,
                    help="instead of the self-play step: search ONE tree for this many seconds with --pending simulations in "
                         "flight (players.AZPlayer's request; one JSON line, metric single_tree_reads_per_sec)")
    ap.add_argument("--pending", type=int, default=64, help="max_pending_evals of --single-tree")
    ap.add_argument("--train-data", type=int, default=0,
                    help="instead of the self-play step: time the training data path (dataset build + batch assembly) "
                         "on this many synthetic replay rows in HBM (one JSON line, metric train_data_rows_per_sec)")
    ap.add_argument("--train-step", type=int, default=0,
                    help="instead of the self-play step: time this many optimizer steps (batch 4096, the --board / --channels / --blocks "
                         "network) with the residual tower on csrc/train.hip and on torch (one JSON line, metric train_samples_per_sec)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.fresh_population:
        args.population = "fresh"
    special = args.full_games > 0 or args.train_data > 0 or args.single_tree > 0 or args.train_step > 0
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not special:
        # before anything touches the GPU: the baseline forks one worker process per host core
        cpu = cpu_baseline(args.board, args.board, args.sims, args.cpu_seconds, args.channels, args.blocks)
    import torch
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):  # launched by torch.distributed.run
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.full_games > 0:
        full_games(args, rank, local_rank, world, torch)
        return
    if args.train_data > 0:
        train_data_bench(args, local_rank, torch)
        return
    if args.train_step > 0:
        train_step_bench(args, local_rank)
        return
    if args.single_tree > 0:
        single_tree_bench(args, local_rank, torch)
        return
    eng, m = run_engine(args, args.precision, args.steps, args.warmup, rank, local_rank, world, dist, torch)
    # replay all-gather at iteration end (multi-GPU): whatever finished + a fixed synthetic shard
    gather = None
    if dist is not None:
        from dotsboxesaz_amd.self_play import gather_replay
        rows_total, ms = gather_replay(eng, dist, synthetic_rows=args.slots * 8)
        gather = dict(rows=rows_total, ms=ms)
    eng.close()
    side = None
    if args.evaluator == "resnet" and args.precision >= 1 and not args.no_f32_side_run:
        st = max(20, args.steps // 6)
        # (exact-f32 steps are 6x longer: the side run keeps the cheap population; it reports the tower's f32 efficiency)
        eng0, m0 = run_engine(args, 0, st, max(5, args.warmup // 5), rank, local_rank, world, dist, torch,
                              population="random" if args.population == "games" else None)
        eng0.close()
        side = (m0, st)
    # the second half of the metric, measured directly: complete games from the empty board on every rank's slots
    leg = None
    n_leg = args.slots if args.games_leg < 0 else args.games_leg
    if n_leg > 0:
        leg = play_complete_games(args, n_leg, args.slots, rank, local_rank, torch, budget_s=args.games_leg_budget)
        if dist is not None:
            t = torch.tensor([float(leg["games"]), leg["seconds"]], dtype=torch.float64).cuda()
            mx = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            leg["games_all_ranks"], leg["seconds_max_over_ranks"] = float(t[0]), float(mx[1])
            leg["value"] = float(t[0]) / float(mx[1])

    if rank == 0:
        rows = cols = args.board
        HW = (rows + 1) * (cols + 1)
        exp, spath, term = m["expansions"], m["sum_path"], m["terminal_leaves"]
        out = {
            "metric": "mcts_node_expansions_per_sec", "value": m["exp_all"] / m["dt_max"], "unit": "expansions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * m["dt_max"] / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == 0 or args.evaluator != "resnet" else
                     "f32 via f16x3 (hi,lo)-split MFMA, f32 accumulate; max |dp|,|dv| vs torch fp32 = 1e-6 (tests/test_hip_nn.py)",
            "data": "synthetic",
            "config": {"workload": "%dx%d board, %d concurrent games/GPU, %d sims/move, evaluator=%s %dx%d "
                                   "random-init, noise (0.8,0.25), tree reuse, mid-game steady-state population"
                                   % (rows, cols, args.slots, args.sims, args.evaluator, args.blocks, args.channels),
                       "baseline_config": "configs[2]" if (args.board, args.slots, args.sims) == (6, 8192, 800) else "custom",
                       "parallelism": "games sharded, %d rank(s)" % world},
            "per_gpu": m["exp_all"] / m["dt_max"] / world,
            "expansions_per_step": m["exp_all"] / max(1, args.steps) / world,
            "step_note": "a step = one simulation for every game whose leaf is evaluated (or needs no network) in it; with the "
                         "network's evaluation list cut back to full rounds of workgroups (DESIGN 4) the slots behind "
                         "the cut complete theirs one step later: expansions_per_step < slots, value counts completed searches only",
            "nn_evals_per_sec": m["evals_all"] / m["dt_max"],
            "mean_path_len": spath / max(1, exp), "terminal_leaf_fraction": term / max(1, exp),
            "cache_hit_fraction": m["cache_hits"] / max(1, exp),
            "pool_high_water": m["pool_high_water"], "nodes_per_slot": m["nodes_per_slot"],
            "moves_played": m["moves_played"], "pool_resets": m["pool_resets"],
            # nn_precision = 1: evaluations whose activations left f16's range and were redone by the exact-f32 tower (timed steps)
            "f32_fallback_evals": m["f32_fallback_evals"], "f32_fallback_fraction": m["f32_fallback_evals"] / max(1, m["nn_evals"]),
        }
        from dotsboxesaz_amd import _lib as _dl
        out["build"] = _dl.build_info()
        # games/s: expansions/s divided by the measured mean expansions of a full game
        # (DESIGN.md "Measurement"; 6x6 @ 800 sims: 63.2k)
        out["population"] = {"games": "mid-game positions reached by quick play (%d reads per move), staggered full searches, %d "
                                      "untimed preparation steps" % (args.quick_reads, m["prep_steps"]),
                             "random": "uniformly random plies, staggered full searches, %d untimed preparation steps" % m["prep_steps"],
                             "fresh": "uniformly random plies, fresh trees"}[args.population]
        if leg:
            out["games_per_sec"] = leg["value"]  # whole-job aggregate, measured on complete games
            out["games_leg"] = leg
        exp_per_game = (leg or {}).get("expansions_per_game") or \
            {(6, 800): 63201.0, (3, 100): 1901.0, (9, 1600): 280481.0}.get((args.board, args.sims))
        if exp_per_game:
            out["games_per_sec_est"] = out["value"] / exp_per_game  # step rate / expansions per game: an estimate
            out["expansions_per_game_assumed"] = exp_per_game
        # tree kernels: HBM roofline with SURVEY 8d's algorithmic bytes per simulation at the measured path length
        A = 2 * HW
        L = spath / max(1, exp)
        bps = (L - 1) * (12 * A + A / 4 + 12) + L * 16 + (12 * A + (A + 7) // 8 + 8) + 2 * (3 * HW * 4) + 4 * A + 4
        if args.evaluator == "resnet" and m["ms_nn_tower"] > 0:
            out["roofline"] = tower_roofline(args, m, args.steps, args.precision)
            t_tree = (m["ms_total"] - m["ms_nn_tower"]) * 1e-3  # tree kernels + head FC
            out["roofline_tree"] = {"bound": "hbm", "kernel": "k_select+k_expand_backup (+k_head_fc in the time)",
                                    "achieved": exp * bps / t_tree / 1e9, "peak": 8000.0, "unit": "GB/s",
                                    "frac": exp * bps / t_tree / 1e9 / 8000.0, "bytes_per_sim": bps,
                                    "note": "latency-bound pointer chase (one wave per game, ~L dependent node "
                                            "visits per sim); a low HBM fraction is expected (SURVEY 8d)"}
            if args.precision == 1 and (args.board, args.channels, args.blocks) == (6, 64, 20):
                tr, note, busy = tower_traffic(PMC_TOWER_PROFILE, out["build"].get("nn"), m["nn_evals"] / args.steps)
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_note"] = note
                if busy is not None:
                    out["roofline"]["mfma_busy_frac"] = busy
            else:
                out["roofline"]["traffic"] = None
        else:
            ach = exp * bps / (m["ms_total"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_select+k_expand_backup", "achieved": ach, "peak": 8000.0,
                               "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "bytes_per_sim": bps}
        if side:
            m0, st = side
            out["exact_f32_mode"] = {"value": m0["exp_all"] / m0["dt_max"], "unit": "expansions/s", "steps": st,
                                     "ms_per_step": 1e3 * m0["dt_max"] / st, "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                                     "roofline": tower_roofline(args, m0, st, 0)}
        if gather:
            out["replay_allgather"] = gather
        if cpu:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Times the optimizer step of the generation loop fed by the HIP batch kernel: ResNetZero 20x64, SGD momentum.  Residual
tower forward/backward on csrc/train.hip (default) or on torch/MIOpen (--torch); bn_input, conv0 and both heads on csrc/train.hip as
well (default: the whole network is two C calls) or on torch (--torch-heads); AlphaZeroLoss and the SGD update are HIP kernels either
way (dbaz_az_loss, dbaz_sgd_step; --torch-loss / --torch-sgd).  Prints one JSON line.

    python tools/train_step_time.py [batch] [--torch] [--torch-heads] [--steps K]   (bench.py --train-step prints both and a roofline)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(batch=4096, hip=True, steps=10, board=6, channels=64, blocks=20, device=0, hip_sgd=True, hip_loss=None, hip_heads=None):
    import torch
    from dotsboxesaz_amd import nn as dnn
    from dotsboxesaz_amd import train as T
    from dotsboxesaz_amd.engine import Engine
    from dotsboxesaz_amd.self_play import _DevBuf
    from dotsboxesaz_amd.train_data import ReplayStore, SymmetriesGenerator
    e = Engine(board, board, 256, mcts_num_read=12, noise=(0.8, 0.25), evaluator="uniform", seed=1, device=device)
    e.selfplay_start(512, 0)
    e.run()
    ptr, n, rb = e.replay_rows_dev()
    rows = torch.as_tensor(_DevBuf(ptr, n * rb), device=torch.device("cuda", device)).view(n, rb).clone()
    store = ReplayStore(e)
    store.add_generation(0, rows)
    ds = store.dataset(train=True, pos_average=True)
    params = dnn.resnet_params(board, board, channels, blocks)
    torch.manual_seed(0)
    model = dnn.ResNetZero(params).cuda(device)
    model.train(True)
    # HipSGD: one HIP launch per step (train.train uses it too)
    opt = (T.HipSGD if hip_sgd else torch.optim.SGD)(model.parameters(), lr=1e-2, momentum=0.9, weight_decay=1e-4)
    crit = T.AlphaZeroLoss()
    sym = SymmetriesGenerator(e)

    host_ms = [0.0, 0.0, 0.0, 0.0]  # host time in forward / loss + backward / optimizer / reading the previous step's scalars

    def epoch(k):
        done, t_data, pend = 0, 0.0, None
        while done < k:
            it = iter(ds.loader(batch, True, True, sym))
            while done < k:
                t0 = time.perf_counter()
                try:
                    boards, pi, z = next(it)
                except StopIteration:
                    break
                t1 = time.perf_counter()
                t_data += t1 - t0
                p, v = T.training_forward(model, boards, hip_tower=hip, hip_heads=hip_heads)
                t2 = time.perf_counter()
                loss, parts = crit.tensors(p, v, pi, z, hip=hip_loss)  # as train.train(): the scalars are read one step late
                loss.backward()
                t3 = time.perf_counter()
                opt.step()
                opt.zero_grad()
                t4 = time.perf_counter()
                if pend is not None:
                    pend[0].item(), pend[1].item()
                t5 = time.perf_counter()
                for i, dt_ in enumerate((t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                    host_ms[i] += 1e3 * dt_
                pend = parts
                done += 1
        torch.cuda.synchronize()
        return t_data

    epoch(3)
    host_ms[:] = [0.0, 0.0, 0.0, 0.0]
    prof = None
    if os.environ.get("TRAIN_STEP_PROFILE"):  # diagnostic: cProfile of the timed steps' host side
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    import gc
    gc.collect()
    gc.freeze()  # as train.train() does: a GC pass over torch's ~1e6 long-lived objects inside a forward pass costs 2-3 ms of launches
    if os.environ.get("TRAIN_STEP_NO_GC"):  # diagnostic
        gc.disable()
    t0 = time.perf_counter()
    td = epoch(steps)
    dt = time.perf_counter() - t0
    if prof is not None:
        import io
        import pstats
        prof.disable()
        buf = io.StringIO()
        pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(14)
        print(buf.getvalue(), file=sys.stderr)
    where = ("whole network on csrc/train.hip" if hip_heads is not False else "residual tower on csrc/train.hip, stem and heads on torch") if hip \
        else "network on torch (MIOpen / rocBLAS)"
    out = {"what": "training step fed by k_make_batch; %s; loss %s, SGD %s" % (where, "torch" if hip_loss is False else "HIP", "HIP" if hip_sgd else "torch"),
           "board": "%dx%d" % (board, board), "net": "ResNetZero %dx%d" % (blocks, channels), "batch": batch, "dataset_rows": len(ds),
           "steps": steps, "ms_per_step": 1e3 * dt / steps, "host_ms_fwd_bwd_opt_read": [round(x / steps, 3) for x in host_ms], "ms_data_per_step": 1e3 * td / steps, "samples_per_sec": batch * steps / dt}
    e.close()
    return out


if __name__ == "__main__":
    pos = [a for a in sys.argv[1:] if not a.startswith("--") and not (sys.argv[sys.argv.index(a) - 1] == "--steps")]
    B = int(pos[0]) if pos else 4096
    K = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
    print(json.dumps(measure(B, "--torch" not in sys.argv, K, hip_sgd="--torch-sgd" not in sys.argv,
                             hip_loss=False if "--torch-loss" in sys.argv else None,
                             hip_heads=False if "--torch-heads" in sys.argv else None)))

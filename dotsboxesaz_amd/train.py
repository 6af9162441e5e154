"""Training step of the generation loop -- host-side mirror of the reference's
`NeuralNetWrapper.train` (nn.py:175-274), `AlphaZeroLoss` (nn.py:131-138), checkpoint format
(nn.py:292-313) and `GenerationLrScheduler` (nn.py:276-289).

What runs where: the DATA side (dataset build from replay rows in HBM, batch gather, symmetries) is
hand-written HIP (`train_data.py` -> csrc/replay.hip).  The NETWORK of a shipped-shape ResNetZero (3 input
planes, 64 channels, 16 head channels) in training mode on the GPU runs forward and backward in hand-written
HIP as well: stem, residual tower and both heads are two C calls (`train_tower.network_forward` ->
csrc/train_net.hip + csrc/train.hip); other head shapes keep the tower in HIP and the stem / heads on torch
(`hip_heads=False` forces that, `hip_tower=False` the whole network on torch -- the CPU goldens' path).
`AlphaZeroLoss` and the SGD update are HIP kernels too (`dbaz_az_loss`, `dbaz_sgd_step`).  torch owns the
parameter / gradient tensors, the autograd graph around those calls, checkpoints and the LR schedule --
plumbing around the product, never on the self-play path: the weight containers of `nn.py` still refuse
`forward()`; `training_forward` composes their sub-modules explicitly where torch evaluates them.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn as tnn


def _get(d, k, default=None):
    if isinstance(d, dict):
        return d.get(k, default)
    return getattr(d, k, default)


def training_forward(model, x, hip_tower=None, hip_heads=None):
    """(log_softmax policy, tanh value) of a `nn.ResNetZero` / `nn.SimpleNN` container, in the
    reference's operation order (nn.py:23-28,48-57,81-86,98-104,117-122; dots_boxes_nn.py:85-98).
    hip_tower: None = the HIP residual tower whenever it applies (training mode, CUDA tensor, 64 channels),
    False = torch's, True = required.  hip_heads: None = bn_input, conv0 and both heads on the HIP kernels too whenever the tower
    is and the model has the shipped stem / head shape (train_tower.net_supported), False = those layers on torch, True = required."""
    if getattr(model, "kind", None) == "simplenn":
        for i in range(5):
            x = getattr(model, "bn%d" % i)(F.relu(getattr(model, "conv%d" % i)(x)))
        x = x.view(x.size(0), -1)
        x = model.bn_fc0(F.relu(model.fc0(x)))
        x = model.bn_fc1(F.relu(model.fc1(x)))
        return F.log_softmax(model.policy_fc(x), dim=1), torch.tanh(model.value_fc(x))
    use_hip = False
    if hip_tower is not False and model.training and x.is_cuda:
        from . import train_tower
        use_hip = train_tower.supported(model, x)
    if hip_tower is True and not use_hip:
        raise RuntimeError("the HIP training tower needs a 64-channel ResNetZero in training mode on a CUDA tensor")
    r, ph, vh = model.resnet, model.policy_head, model.value_head
    if use_hip and hip_heads is not False and train_tower.net_supported(model, x):
        # the whole network -- stem, residual tower, heads -- forward and backward on csrc/train.hip (two C calls)
        return train_tower.network_forward(model, x)
    if hip_heads is True:
        raise RuntimeError("the HIP stem / heads need the shipped ResNetZero shape (3 input planes, 16 head channels) in training mode")
    if use_hip:
        # residual blocks and every BatchNorm2d (+ its ReLU) on csrc/train.hip; convs 3->64 / 1x1 and the FCs on torch
        bn = train_tower.batch_norm_train
        x = bn(model.bn_input, x)
        x = bn(r.bn0, r.conv0(x), relu=True)
        x = train_tower.resblocks_forward(model, x)
        p = bn(ph.bn0, ph.conv0(x), relu=True)
        v = bn(vh.bn0, vh.conv0(x), relu=True)
    else:
        x = model.bn_input(x)
        x = F.relu(r.bn0(r.conv0(x)))
        for blk in r.resblocks:
            y = F.relu(blk.bn1(blk.conv1(x)))
            y = blk.bn2(blk.conv2(y))
            y += x
            x = F.relu(y)
        p = F.relu(ph.bn0(ph.conv0(x)))
        v = F.relu(vh.bn0(vh.conv0(x)))
    p = F.log_softmax(ph.fc(p.view(p.size(0), -1)), dim=1)
    v = F.relu(vh.fc0(v.view(v.size(0), -1)))
    v = torch.tanh(vh.fc1(v))
    return p, v


class _AzLossFn(torch.autograd.Function):
    """AlphaZeroLoss forward + backward in two HIP launches (dbaz_az_loss, csrc/train.hip): returns the 3-vector
    (loss_v + loss_pi, loss_pi, loss_v); gradients flow to the policy log-probabilities and the value."""

    @staticmethod
    def forward(ctx, p, v, pi, z):
        import ctypes as C
        from . import _lib
        L = _lib.load()
        pc, vc = p.detach().contiguous().float(), v.detach().contiguous().float()
        pic, zc = pi.detach().contiguous().float(), z.detach().contiguous().float()
        n, A = pc.shape
        loss3 = torch.empty(3, dtype=torch.float32, device=pc.device)
        d_p, d_v = torch.empty_like(pc), torch.empty_like(vc)
        ws = torch.empty(int(L.dbaz_az_loss_workspace_bytes()) // 8, dtype=torch.float64, device=pc.device)
        with torch.cuda.device(pc.device):
            rc = L.dbaz_az_loss(pc.data_ptr(), vc.data_ptr(), pic.data_ptr(), zc.data_ptr(), n, A, 1.0, loss3.data_ptr(), d_p.data_ptr(),
                                d_v.data_ptr(), ws.data_ptr(), C.c_void_p(torch.cuda.current_stream(pc.device).cuda_stream))
        if rc != _lib.OK:
            raise RuntimeError((L.dbaz_trainer_last_error(None) or b"dbaz_az_loss failed").decode())
        ctx.save_for_backward(d_p, d_v)
        ctx.v_shape = v.shape
        return loss3

    @staticmethod
    def backward(ctx, g3):
        d_p, d_v = ctx.saved_tensors
        g = g3[0]  # only the total carries a gradient (loss_pi / loss_v are read out detached)
        return d_p * g, (d_v * g).view(ctx.v_shape), None, None


class AlphaZeroLoss(tnn.Module):
    """nn.py:131-138: mean squared value error + mean cross entropy against the MCTS policy;
    returns (loss, (loss_pi, loss_v) as python floats).  On CUDA float32 tensors the forward and backward run as HIP kernels
    (dbaz_az_loss); elsewhere (the CPU goldens of tests/golden/train.npz) in torch, statement for statement as the reference."""

    def forward(self, p, v, pi, z):
        loss, (loss_pi, loss_v) = self.tensors(p, v, pi, z)
        return loss, (loss_pi.item(), loss_v.item())

    @staticmethod
    def tensors(p, v, pi, z, hip=None):
        """The same without the two host synchronisations of `.item()`: (loss, (loss_pi, loss_v)) as 0-d tensors.
        hip: None = the HIP kernels whenever the tensors are CUDA float32 [n, A] / [n, 1]; False = torch."""
        use = hip is not False and p.is_cuda and p.dtype == torch.float32 and p.dim() == 2 and v.numel() == p.shape[0] \
            and z.numel() == p.shape[0] and pi.shape == p.shape
        if hip is True and not use:
            raise RuntimeError("the HIP loss needs CUDA float32 tensors p, pi [n, A] and v, z [n, 1]")
        if use:
            l3 = _AzLossFn.apply(p, v, pi, z)
            return l3[0], (l3[1].detach(), l3[2].detach())
        loss_v = (z - v).pow(2).mean()
        loss_pi = -(pi * p).sum(1).mean()
        return loss_v + loss_pi, (loss_pi.detach(), loss_v.detach())


class HipSGD(torch.optim.SGD):
    """torch.optim.SGD (momentum, weight decay; dampening 0, no nesterov: what the reference configures, nn.py:179 with
    configuration.py:62-66) whose step() is ONE HIP launch over all parameter tensors (dbaz_sgd_step, csrc/train.hip) instead of
    torch's three or four multi-tensor launches (the tensors' pointers travel as kernel arguments).  It IS a torch SGD: param_groups, state[p]["momentum_buffer"], state_dict() and
    load_state_dict() are torch's, so the reference's checkpoints ({'optimizer_dict': optimizer.state_dict()}, nn.py:292-313)
    load and save unchanged.  Anything the kernel does not cover (CPU parameters, nesterov, dampening, maximize, sparse or
    non-float32 gradients) goes through torch's own step()."""

    def _plan(self, params):
        """Chunk tables, device scratch and the (stable) parameter pointers for this set of tensors; rebuilt only when the set
        changes.  The per-step path below allocates no Python containers beyond two lists: a step that creates hundreds of tuples
        makes the cyclic garbage collector run inside the next forward pass (measured: +2.3 ms per step)."""
        old = getattr(self, "_plan_params", None)
        if old is not None and len(old) == len(params):
            same = True
            for i in range(len(params)):
                if old[i] is not params[i]:
                    same = False
                    break
            if same and self._plan_ptr0 == params[0].data_ptr():
                return
        import ctypes as C
        import numpy as np
        dev = params[0].device
        ct, co = [], []
        for t, p in enumerate(params):
            nc = (p.numel() + 2047) // 2048
            ct.append(np.full(nc, t, np.int32))
            co.append(np.arange(nc, dtype=np.int32))
        self._chunk_tensor = torch.from_numpy(np.concatenate(ct)).to(dev)
        self._chunk_off = torch.from_numpy(np.concatenate(co)).to(dev)
        self._n_chunks = int(self._chunk_tensor.numel())
        n = len(params)
        self._table = torch.empty((n, 4), dtype=torch.int64, device=dev)   # device scratch of dbaz_sgd_step
        self._numels = (C.c_int64 * n)(*[p.numel() for p in params])
        self._p_ptrs = (C.c_void_p * n)(*[p.data_ptr() for p in params])
        self._g_ptrs = (C.c_void_p * n)()
        self._b_ptrs = (C.c_void_p * n)()
        self._plan_params = list(params)
        self._plan_ptr0 = params[0].data_ptr()

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes as C
        from . import _lib
        if closure is not None or len(self.param_groups) != 1:
            return super().step(closure)
        g = self.param_groups[0]
        params = [p for p in g["params"] if p.grad is not None]
        ok = bool(params) and not g["nesterov"] and g["dampening"] == 0 and not g.get("maximize", False)
        if ok:
            dev = params[0].device
            for p in params:
                gr = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and gr.dtype == torch.float32 and not gr.is_sparse
                        and gr.is_contiguous() and p.device == dev):
                    ok = False
                    break
        if not ok:
            return super().step()
        self._plan(params)
        mom = float(g["momentum"])
        n = len(params)
        gp, bp = self._g_ptrs, self._b_ptrs
        for i in range(n):
            p = params[i]
            gp[i] = p.grad.data_ptr()      # (new tensors every step: zero_grad sets them to None)
            if mom != 0.0:                 # (without momentum torch keeps no state: neither do we)
                st = self.state[p]
                b = st.get("momentum_buffer")
                if b is None:
                    b = st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.preserve_format)  # == torch's first step (buf = d)
                bp[i] = b.data_ptr()
        L = _lib.load()
        with torch.cuda.device(dev):
            rc = L.dbaz_sgd_step(n, self._p_ptrs, gp, bp if mom != 0.0 else None, self._numels, self._table.data_ptr(),
                                 self._chunk_tensor.data_ptr(), self._chunk_off.data_ptr(), self._n_chunks, float(g["lr"]), mom,
                                 float(g["weight_decay"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != _lib.OK:
            raise RuntimeError((L.dbaz_trainer_last_error(None) or b"dbaz_sgd_step failed").decode())
        return None


class GenerationLrScheduler:
    """nn.py:276-289: the learning rate of the last schedule key <= generation."""

    def __init__(self, schedule):
        assert schedule is not None
        self.schedule = schedule

    def __call__(self, generation):
        lr = None
        for g in range(generation + 1):
            if g in self.schedule:
                lr = self.schedule[g]
        assert lr is not None
        return lr

    def __repr__(self):
        return "GenerationLrScheduler(%s)" % (self.schedule,)


def save_checkpoint(filename, model, optimizer, last_batch_idx):
    """nn.py:292-295: {'last_batch_idx', 'model_dict', 'optimizer_dict'}."""
    torch.save({"last_batch_idx": last_batch_idx, "model_dict": model.state_dict(),
                "optimizer_dict": optimizer.state_dict()}, filename)


def load_checkpoint(filename, model, optimizer, to_device):
    """nn.py:296-313 (missing file -> ValueError).  Checkpoints are tensors and plain containers:
    loaded with weights_only=True."""
    if not os.path.isfile(filename):
        raise ValueError("=> no checkpoint found at '%s'" % filename)
    ck = torch.load(filename, map_location="cpu", weights_only=True)
    model.load_state_dict(ck["model_dict"])
    optimizer.load_state_dict(ck["optimizer_dict"])
    model.to(to_device)
    for state in optimizer.state.values():
        for k, v in state.items():
            if isinstance(v, torch.Tensor):
                state[k] = v.to(to_device)
    return ck["last_batch_idx"]


def _accuracy(v, z, threshold=0.5, lazy=False):
    with torch.no_grad():
        correct = z.sign().eq(v.sign())
        correct = correct * (v - z).abs().lt(threshold)
        return (correct.sum() if lazy else correct.sum().item()), z.size()[0]


def _batches(dataset, batch_size, shuffle, symmetries, device):
    """Device fast path when the dataset offers it (train_data.ReplayDataset.loader: ONE HIP kernel
    per batch gathers, normalises and transforms in HBM); otherwise the reference's own flow: a torch
    DataLoader over a host dataset, transfer, then `symmetries(boards, pi)` (nn.py:186-216)."""
    if hasattr(dataset, "loader"):
        for b in dataset.loader(batch_size, shuffle=shuffle, drop_last=True, symmetries=symmetries):
            yield b
        return
    from torch.utils import data
    for boards, pi, z in data.DataLoader(dataset, batch_size, shuffle=shuffle, drop_last=True):
        boards, pi, z = boards.to(device), pi.to(device), z.to(device)
        if symmetries is not None:
            boards, pi = symmetries(boards, pi)
        yield boards, pi, z


def train(model, params, train_dataset, val_dataset, writer, generation, device=None, eval_forward=None):
    """NeuralNetWrapper.train (nn.py:175-274): SGD(lr, **optimizer_params), resume from the checkpoint
    of generation-1 when generation > 0, min(2*generation, nb_epochs) epochs of shuffled drop_last
    batches with a random symmetry each, validation pass per epoch, tensorboard-style scalars,
    checkpoint of `generation` written at the end.  Returns the last batch index.
    eval_forward (optional): callable(model) -> callable(boards) -> (log p, v) for the validation passes; NeuralNetWrapper
    passes the HIP inference engine (the fused tower of csrc/nn.hip: 4 096 positions in 2 ms; torch's evaluation-mode forward
    of the same batch takes 56 ms, a third of a generation's training time)."""
    nnp = _get(params, "nn")
    tp = _get(nnp, "train_params")
    device = torch.device(device if device is not None else
                          (_get(nnp, "pytorch_device", "cuda:0") if torch.cuda.is_available() else "cpu"))
    symmetries = _get(tp, "symmetries")
    model.to(device)
    criterion = AlphaZeroLoss()
    # (HipSGD is torch.optim.SGD with a one-launch HIP step() on CUDA parameters; on the CPU it is torch's SGD)
    optimizer = HipSGD(model.parameters(), lr=_get(tp, "lr"), **dict(_get(tp, "optimizer_params") or {}))
    batch_i = 0
    if generation > 0:
        batch_i = load_checkpoint(_get(nnp, "chkpts_filename").format(generation - 1), model, optimizer, device)
    writer.add_scalar("lr", _get(tp, "lr"), batch_i)
    # The step is a few hundred kernel launches issued by Python; a cyclic-GC pass over the process's long-lived objects (torch's
    # modules: ~1e6 of them) in the middle of a forward pass costs 2-3 ms of launch time per step (measured with the profiler:
    # tools/train_step_time.py).  Freeze what exists now into the permanent generation: collections during training then only look
    # at the objects the loop itself creates.
    import gc
    gc.collect()
    gc.freeze()
    try:
        batch_i = _train_epochs(model, tp, nnp, train_dataset, val_dataset, writer, generation, device, eval_forward, criterion, optimizer,
                                symmetries, batch_i)
    finally:
        gc.unfreeze()
    save_checkpoint(_get(nnp, "chkpts_filename").format(generation), model, optimizer, batch_i)
    return batch_i


def _train_epochs(model, tp, nnp, train_dataset, val_dataset, writer, generation, device, eval_forward, criterion, optimizer, symmetries,
                  batch_i):
    for epoch in range(min(2 * generation, _get(tp, "nb_epochs"))):
        model.train(True)
        tr_loss, tr_batches, tr_ok, tr_tot = 0, 0, 0, 1
        # The reference reads loss_pi / loss_v / the accuracy count back with .item() in every step (nn.py:136-138,180-184), i.e.
        # it drains the GPU queue three times per batch.  The values and the order of the writer calls are the same here, but a
        # step's scalars are fetched only after the NEXT step has been queued, so the device never waits for the host.
        pending = None

        def flush(pend):
            nonlocal tr_ok, tr_loss
            if pend is None:
                return
            bi, lpi_t, lv_t, c_t = pend
            lpi, lv = lpi_t.item(), lv_t.item()
            tr_ok += int(c_t.item())
            tr_loss += lpi + lv
            writer.add_scalars("loss", {"pi/train": lpi, "v/train": lv, "total/train": lpi + lv}, bi)

        for boards, pi, z in _batches(train_dataset, _get(tp, "train_batch_size"), True, symmetries, device):
            batch_i += 1
            tr_batches += 1
            p, v = training_forward(model, boards)
            loss, (lpi_t, lv_t) = criterion.tensors(p, v, pi, z)
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            c_t, t = _accuracy(v, z, lazy=True)
            tr_tot += t
            flush(pending)
            pending = (batch_i, lpi_t, lv_t, c_t)
        flush(pending)
        val_loss, loss_v, loss_pi, val_ok, val_tot = 0.0, 0.0, 0.0, 0, 1
        if val_dataset:
            model.train(False)
            val_batches = 0
            fwd = eval_forward(model) if eval_forward is not None else (lambda b: training_forward(model, b))
            for boards, pi, z in _batches(val_dataset, _get(tp, "val_batch_size"), False, symmetries, device):
                val_batches += 1
                with torch.no_grad():
                    p, v = fwd(boards)
                c, t = _accuracy(v, z)
                val_ok += c
                val_tot += t
                _, (_lpi, _lv) = criterion(p, v, pi, z)
                loss_v += _lv
                loss_pi += _lpi
            if val_batches > 0:
                loss_v /= val_batches
                loss_pi /= val_batches
                val_loss = loss_v + loss_pi
                writer.add_scalars("loss", {"pi/eval": loss_pi, "v/eval": loss_v, "total/eval": val_loss}, batch_i)
        writer.add_scalars("accuracy", {"v/train": tr_ok / tr_tot, "v/eval": val_ok / val_tot}, batch_i)
        writer.add_scalar("generation", generation, batch_i)
        print("Epoch %d, train loss= %5f, validation loss= %5f" % (epoch, tr_loss / max(1, tr_batches), val_loss), flush=True)
    return batch_i


def window_where(generation):
    """coach.py:148-149: generations kept in the training window (`generation >= max(0, g - ws)`)."""
    ws = max(4, min(4 + (generation - 4) // 2, 20))
    return max(0, generation - ws)

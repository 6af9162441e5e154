"""Training-mode forward/backward of ResNetZero's residual blocks on the hand-written HIP kernels of csrc/train.hip
(C ABI: dbaz_trainer_* in include/dbaz.h), exposed to torch autograd.

Reference: the `for block in self.resblocks` loop of ResNet.forward (nn.py:23-28) with ResBlock.forward (nn.py:48-57)
under `model.train(True)` in NeuralNetWrapper.train (nn.py:203-221).  torch is plumbing here: it owns the parameter and
gradient tensors and calls the C ABI with their device pointers on its current stream; the arithmetic of the blocks --
conv3x3, batch statistics, normalisation, ReLU, skip connections and all their gradients -- runs in train.hip.
"""
import ctypes as C
import weakref

import torch

from . import _lib


class TrainerError(RuntimeError):
    pass


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class TowerTrainer:
    """One handle per (board, blocks, max batch, device); holds the activations of one forward pass."""

    def __init__(self, rows, cols, channels, blocks, max_batch, device=0):
        self._L = _lib.load()
        h = C.c_void_p()
        rc = self._L.dbaz_trainer_create(rows, cols, channels, blocks, max_batch, device, C.byref(h))
        if rc != _lib.OK:
            msg = self._L.dbaz_trainer_last_error(None)
            raise TrainerError(msg.decode() if msg else "dbaz_trainer_create failed (%d)" % rc)
        self.h = h
        global handles_created
        handles_created += 1
        self.rows, self.cols, self.channels, self.blocks, self.max_batch, self.device = rows, cols, channels, blocks, max_batch, device
        self.generation = 0  # id of the forward pass whose activations the handle holds (a handle holds ONE)

    def _ck(self, rc):
        if rc != _lib.OK:
            msg = self._L.dbaz_trainer_last_error(self.h)
            raise TrainerError(msg.decode() if msg else "error %d" % rc)

    def close(self):
        if getattr(self, "h", None):
            self._L.dbaz_trainer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, x, conv_w, conv_b, bn_w, bn_b, run_mean, run_var):
        """Returns (out, generation): the handle now holds THIS pass's activations; an older pass can no longer be differentiated."""
        if x.shape[0] > self.max_batch:
            raise TrainerError("batch %d exceeds the handle's max_batch %d" % (x.shape[0], self.max_batch))
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):  # the C calls launch on the tensors' device, whatever torch's current device is
            stream = torch.cuda.current_stream(x.device).cuda_stream
            self._ck(self._L.dbaz_trainer_forward(self.h, x.shape[0], x.data_ptr(), _ptr_array(conv_w), _ptr_array(conv_b),
                                                  _ptr_array(bn_w), _ptr_array(bn_b), _ptr_array(run_mean), _ptr_array(run_var),
                                                  out.data_ptr(), C.c_void_p(stream)))
        self.generation += 1
        return out, self.generation

    def backward(self, grad_out, bn_w, g_conv_w, g_conv_b, g_bn_w, g_bn_b, generation=None):
        if generation is not None and generation != self.generation:
            raise TrainerError("backward of forward pass #%d, but the trainer handle now holds pass #%d: a handle keeps the activations "
                               "of ONE forward pass (two micro-batches in one graph, or two graphs alive at once, need one "
                               "TowerTrainer each -- train_tower.resblocks_forward(..., trainer=...))" % (generation, self.generation))
        gx = torch.empty_like(grad_out)
        with torch.cuda.device(grad_out.device):
            stream = torch.cuda.current_stream(grad_out.device).cuda_stream
            self._ck(self._L.dbaz_trainer_backward(self.h, grad_out.data_ptr(), _ptr_array(bn_w), gx.data_ptr(), _ptr_array(g_conv_w),
                                                   _ptr_array(g_conv_b), _ptr_array(g_bn_w), _ptr_array(g_bn_b), C.c_void_p(stream)))
        return gx


    # ---- the whole network (stem, tower, heads): dbaz_trainer_net_forward / _backward
    def net_forward(self, x, tensors, running, head_channels, n_actions, value_fc):
        """x [n,3,H,W] -> (logp [n,A], v [n,1], generation).  tensors / running: _NetArgs (keep them alive until the call returns)."""
        n = x.shape[0]
        if n > self.max_batch:
            raise TrainerError("batch %d exceeds the handle's max_batch %d" % (n, self.max_batch))
        logp = torch.empty((n, n_actions), dtype=torch.float32, device=x.device)
        v = torch.empty((n, 1), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            self._ck(self._L.dbaz_trainer_net_forward(self.h, n, x.data_ptr(), C.byref(tensors.struct), C.byref(running.struct), head_channels,
                                                      n_actions, value_fc, logp.data_ptr(), v.data_ptr(), C.c_void_p(stream)))
        self.generation += 1
        return logp, v, self.generation

    def net_backward(self, x, d_logp, d_v, tensors, grads, generation=None):
        if generation is not None and generation != self.generation:
            raise TrainerError("backward of forward pass #%d, but the trainer handle now holds pass #%d: a handle keeps the activations "
                               "of ONE forward pass" % (generation, self.generation))
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            self._ck(self._L.dbaz_trainer_net_backward(self.h, x.data_ptr(), d_logp.data_ptr(), d_v.data_ptr(), C.byref(tensors.struct),
                                                       C.byref(grads.struct), C.c_void_p(stream)))


class _TowerFn(torch.autograd.Function):
    """autograd node around the two C calls.  args: trainer, running stats (lists), x, then 4 tensors per layer
    (conv.weight, conv.bias, bn.weight, bn.bias) in layer order."""

    @staticmethod
    def forward(ctx, trainer, run_mean, run_var, x, *params):
        L = len(params) // 4
        cw, cb, bw, bb = ([params[4 * l + k].detach().contiguous() for l in range(L)] for k in range(4))
        xc = x.detach().contiguous().float()
        out, gen = trainer.forward(xc, cw, cb, bw, bb, run_mean, run_var)
        ctx.trainer, ctx.bw, ctx.shapes, ctx.generation = trainer, bw, [p.shape for p in params], gen
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = len(ctx.shapes) // 4
        dev = grad_out.device
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.shapes]
        gx = ctx.trainer.backward(grad_out.contiguous().float(), ctx.bw, [grads[4 * l] for l in range(L)], [grads[4 * l + 1] for l in range(L)],
                                  [grads[4 * l + 2] for l in range(L)], [grads[4 * l + 3] for l in range(L)], generation=ctx.generation)
        return (None, None, None, gx) + tuple(grads)


# model -> {(H, W, blocks, device): TowerTrainer}: one handle per MODEL (two models of one shape never share activations), held
# through a WEAK reference to the model -- coach.train_nn builds a new model every generation (coach.py:71), and a handle is ~5 GB
# of HBM at batch 4 096: keyed by id(model) the handles of dead models piled up; now a handle goes with its model (an autograd graph
# that still needs it keeps it alive through ctx.trainer)
_trainers = weakref.WeakKeyDictionary()
handles_created = 0  # TowerTrainer handles made so far (tests)


def _cached_trainer(model, H, W, blocks, device, batch):
    per = _trainers.get(model)
    if per is None:
        per = {}
        _trainers[model] = per
    key = (H, W, blocks, device)
    tr = per.get(key)
    if tr is None or tr.max_batch < batch:
        # (a smaller handle is NOT closed here: an autograd graph may still hold it; it is freed with its last reference)
        tr = TowerTrainer(H - 1, W - 1, 64, blocks, max(int(batch), 1), device)
        per[key] = tr
    return tr


def _bn_ok(bn):
    # csrc/train.hip hard-codes torch's defaults: eps 1e-5, momentum 0.1, affine, running statistics, float32
    return (bn.eps == 1e-5 and bn.momentum == 0.1 and bn.affine and bn.track_running_stats and bn.weight.dtype == torch.float32
            and bn.running_mean is not None and bn.running_mean.dtype == torch.float32)


def _conv_ok(conv, ch):
    return (conv.in_channels == ch and conv.out_channels == ch and tuple(conv.kernel_size) == (3, 3) and tuple(conv.padding) == (1, 1)
            and tuple(conv.stride) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is not None
            and conv.weight.dtype == torch.float32)


def supported(model, x):
    """The HIP tower handles ResNetZero containers with 64 channels on a float32 CUDA/HIP tensor whose blocks are exactly what
    csrc/train.hip computes: 3x3 pad-1 convs with bias, BatchNorm2d with torch's default eps / momentum, affine, running statistics.
    Anything else stays on torch (train.training_forward)."""
    r = getattr(model, "resnet", None)
    if not (x.is_cuda and x.dtype == torch.float32 and r is not None and len(r.resblocks) > 0 and r.conv0.out_channels == 64):
        return False
    # the BatchNorm2d layers outside the blocks run on dbaz_bn2d_* (any eps / momentum, but a number); a bare block stack has none
    outer = [getattr(model, "bn_input", None), getattr(r, "bn0", None), getattr(getattr(model, "policy_head", None), "bn0", None),
             getattr(getattr(model, "value_head", None), "bn0", None)]
    outer = [b for b in outer if b is not None]
    if not all(b.affine and b.track_running_stats and b.momentum is not None and b.weight.dtype == torch.float32 for b in outer):
        return False
    return all(_conv_ok(c, 64) and _bn_ok(b) for blk in r.resblocks for c, b in ((blk.conv1, blk.bn1), (blk.conv2, blk.bn2)))


def resblocks_forward(model, x, trainer=None):
    """x -> the output of model.resnet.resblocks in training mode (batch statistics; running stats and
    num_batches_tracked updated), differentiable.  A trainer handle holds the activations of ONE forward pass: a second
    forward through the same handle before the first one's backward makes that backward raise (TrainerError).  Pass your own
    `trainer` (TowerTrainer) per live graph when several forward passes must be differentiated (micro-batches)."""
    r = model.resnet
    blocks = list(r.resblocks)
    H, W = x.shape[2], x.shape[3]
    tr = trainer
    if tr is None:
        tr = _cached_trainer(model, H, W, len(blocks), x.device.index or 0, x.shape[0])
    params, rm, rv, nbt = [], [], [], []
    for b in blocks:
        for conv, bn in ((b.conv1, b.bn1), (b.conv2, b.bn2)):
            params += [conv.weight, conv.bias, bn.weight, bn.bias]
            rm.append(bn.running_mean)
            rv.append(bn.running_var)
            nbt.append(bn.num_batches_tracked)
    torch._foreach_add_(nbt, 1)   # one launch for the 2*blocks counters (BatchNorm2d.forward: num_batches_tracked += 1)
    return _TowerFn.apply(tr, rm, rv, x, *params)


# ---- training-mode BatchNorm2d (+ ReLU) of the layers outside the residual blocks (bn_input, bn0, the heads' bn0)
class _BNFn(torch.autograd.Function):
    """y = relu?(batch_norm(x)) with batch statistics on csrc/train.hip (dbaz_bn2d_*); running statistics updated in place."""

    @staticmethod
    def forward(ctx, x, weight, bias, run_mean, run_var, eps, momentum, relu):
        L = _lib.load()
        xc = x.detach().contiguous().float()
        n, ch = xc.shape[0], xc.shape[1]
        hw = xc.numel() // (n * ch)
        out = torch.empty_like(xc)
        mean = torch.empty(ch, dtype=torch.float32, device=xc.device)
        invstd = torch.empty_like(mean)
        ws = torch.empty(int(L.dbaz_bn2d_workspace_bytes(ch)) // 8, dtype=torch.float64, device=xc.device)
        w, b = weight.detach().contiguous(), bias.detach().contiguous()
        with torch.cuda.device(xc.device):  # dbaz_bn2d_* launch on the CURRENT device's stream: make it the tensor's
            stream = C.c_void_p(torch.cuda.current_stream(xc.device).cuda_stream)
            rc = L.dbaz_bn2d_forward(xc.data_ptr(), n, ch, hw, w.data_ptr(), b.data_ptr(), run_mean.data_ptr(), run_var.data_ptr(),
                                     float(eps), float(momentum), int(bool(relu)), out.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                     ws.data_ptr(), stream)
        if rc != _lib.OK:
            raise TrainerError((L.dbaz_trainer_last_error(None) or b"dbaz_bn2d_forward failed").decode())
        ctx.save_for_backward(xc, out, w, mean, invstd)
        ctx.relu, ctx.ws = bool(relu), ws
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.load()
        xc, out, w, mean, invstd = ctx.saved_tensors
        n, ch = xc.shape[0], xc.shape[1]
        hw = xc.numel() // (n * ch)
        d = dout.contiguous().float()
        dx, dw, db = torch.empty_like(xc), torch.empty_like(w), torch.empty_like(w)
        with torch.cuda.device(xc.device):
            stream = C.c_void_p(torch.cuda.current_stream(xc.device).cuda_stream)
            rc = L.dbaz_bn2d_backward(d.data_ptr(), out.data_ptr(), xc.data_ptr(), n, ch, hw, w.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                      int(ctx.relu), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), ctx.ws.data_ptr(), stream)
        if rc != _lib.OK:
            raise TrainerError((L.dbaz_trainer_last_error(None) or b"dbaz_bn2d_backward failed").decode())
        return dx, dw, db, None, None, None, None, None


def batch_norm_train(bn, x, relu=False):
    """torch.nn.BatchNorm2d `bn` in training mode on a CUDA tensor (+ ReLU), on the HIP kernels: F.relu(bn(x)) of nn.py:26,82,99
    and bn_input(x) of nn.py:117."""
    bn.num_batches_tracked += 1
    return _BNFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, relu)


# ---- the whole network of the optimizer step on csrc/train_net.hip + csrc/train.hip: p, v = model(boards) (nn.py:108-122) and its backward
_NET_SINGLE = ("bn_input_w", "bn_input_b", "conv0_w", "conv0_b", "bn0_w", "bn0_b", "ph_conv_w", "ph_conv_b", "ph_bn_w", "ph_bn_b", "ph_fc_w",
               "ph_fc_b", "vh_conv_w", "vh_conv_b", "vh_bn_w", "vh_bn_b", "vh_fc0_w", "vh_fc0_b", "vh_fc1_w", "vh_fc1_b")
_NET_BLOCK = ("blk_conv_w", "blk_conv_b", "blk_bn_w", "blk_bn_b")


def _net_parameters(model):
    """The parameters in dbaz_net_tensors order: the 20 single tensors, then per layer (conv.weight, conv.bias, bn.weight, bn.bias)."""
    r, ph, vh = model.resnet, model.policy_head, model.value_head
    single = [model.bn_input.weight, model.bn_input.bias, r.conv0.weight, r.conv0.bias, r.bn0.weight, r.bn0.bias,
              ph.conv0.weight, ph.conv0.bias, ph.bn0.weight, ph.bn0.bias, ph.fc.weight, ph.fc.bias,
              vh.conv0.weight, vh.conv0.bias, vh.bn0.weight, vh.bn0.bias, vh.fc0.weight, vh.fc0.bias, vh.fc1.weight, vh.fc1.bias]
    layers = []
    for b in r.resblocks:
        for conv, bn in ((b.conv1, b.bn1), (b.conv2, b.bn2)):
            layers += [conv.weight, conv.bias, bn.weight, bn.bias]
    return single, layers


class _NetArgs:
    """A dbaz_net_tensors over a list of tensors (keeps the pointer arrays alive)."""

    def __init__(self, single, layers):
        st = _lib.NetTensors()
        for name, t in zip(_NET_SINGLE, single):
            setattr(st, name, t.data_ptr())
        L = len(layers) // 4
        self._arrays = [_ptr_array([layers[4 * l + k] for l in range(L)]) for k in range(4)]
        for name, arr in zip(_NET_BLOCK, self._arrays):
            setattr(st, name, C.cast(arr, C.c_void_p))
        self.struct = st


class _NetRunning:
    def __init__(self, model):
        r, ph, vh = model.resnet, model.policy_head, model.value_head
        st = _lib.NetRunning()
        for name, t in (("bn_input_mean", model.bn_input.running_mean), ("bn_input_var", model.bn_input.running_var),
                        ("bn0_mean", r.bn0.running_mean), ("bn0_var", r.bn0.running_var), ("ph_mean", ph.bn0.running_mean),
                        ("ph_var", ph.bn0.running_var), ("vh_mean", vh.bn0.running_mean), ("vh_var", vh.bn0.running_var)):
            setattr(st, name, t.data_ptr())
        bns = [bn for b in r.resblocks for bn in (b.bn1, b.bn2)]
        self._arrays = [_ptr_array([bn.running_mean for bn in bns]), _ptr_array([bn.running_var for bn in bns])]
        st.blk_mean = C.cast(self._arrays[0], C.c_void_p)
        st.blk_var = C.cast(self._arrays[1], C.c_void_p)
        self.struct = st
        self.counters = [model.bn_input.num_batches_tracked, r.bn0.num_batches_tracked, ph.bn0.num_batches_tracked,
                         vh.bn0.num_batches_tracked] + [bn.num_batches_tracked for bn in bns]


class _NetFn(torch.autograd.Function):
    """autograd node around dbaz_trainer_net_forward / _backward.  args: trainer, running statistics, geometry, x, then the
    parameters in _net_parameters order."""

    @staticmethod
    def forward(ctx, trainer, running, geo, x, *params):
        ps = [p.detach().contiguous() for p in params]
        tensors = _NetArgs(ps[:len(_NET_SINGLE)], ps[len(_NET_SINGLE):])
        xc = x.detach().contiguous().float()
        logp, v, gen = trainer.net_forward(xc, tensors, running, *geo)
        ctx.trainer, ctx.tensors, ctx.ps, ctx.x, ctx.generation = trainer, tensors, ps, xc, gen
        return logp, v

    @staticmethod
    def backward(ctx, d_logp, d_v):
        dev = ctx.x.device
        grads = [torch.empty(p.shape, dtype=torch.float32, device=dev) for p in ctx.ps]
        g = _NetArgs(grads[:len(_NET_SINGLE)], grads[len(_NET_SINGLE):])
        ctx.trainer.net_backward(ctx.x, d_logp.contiguous().float(), d_v.contiguous().float(), ctx.tensors, g, generation=ctx.generation)
        return (None, None, None, None) + tuple(grads)


def net_supported(model, x):
    """The whole-network HIP pass covers what `supported` covers plus the shipped stem and heads: 3 input planes, 16 head channels,
    1x1 head convs with bias, the heads' fc layers with bias, BatchNorm2d with torch's defaults everywhere."""
    if not supported(model, x):
        return False
    r, ph, vh = model.resnet, getattr(model, "policy_head", None), getattr(model, "value_head", None)
    bn_in = getattr(model, "bn_input", None)
    if ph is None or vh is None or bn_in is None or x.dim() != 4 or x.shape[1] != 3:
        return False
    H, W = x.shape[2], x.shape[3]

    def conv1x1(c):
        return (c.in_channels == 64 and c.out_channels == 16 and tuple(c.kernel_size) == (1, 1) and tuple(c.stride) == (1, 1)
                and tuple(c.padding) == (0, 0) and c.groups == 1 and c.bias is not None and c.weight.dtype == torch.float32)

    c0 = r.conv0
    stem = (c0.in_channels == 3 and tuple(c0.kernel_size) == (3, 3) and tuple(c0.padding) == (1, 1) and tuple(c0.stride) == (1, 1)
            and tuple(c0.dilation) == (1, 1) and c0.groups == 1 and c0.bias is not None and bn_in.num_features == 3)
    fcs = (ph.fc.in_features == 16 * H * W and vh.fc0.in_features == 16 * H * W and vh.fc1.in_features == vh.fc0.out_features
           and vh.fc1.out_features == 1 and ph.fc.bias is not None and vh.fc0.bias is not None and vh.fc1.bias is not None
           and ph.fc.out_features <= 1024 and vh.fc0.out_features <= 256)
    return bool(stem and conv1x1(ph.conv0) and conv1x1(vh.conv0) and fcs and all(_bn_ok(b) for b in (bn_in, r.bn0, ph.bn0, vh.bn0)))


def network_forward(model, x, trainer=None):
    """(log_softmax policy [n, A], tanh value [n, 1]) of a ResNetZero in training mode, every layer on csrc/train.hip; differentiable
    with respect to all parameters (not x).  One handle holds ONE forward pass (see resblocks_forward)."""
    r = model.resnet
    H, W = x.shape[2], x.shape[3]
    tr = trainer
    if tr is None:
        tr = _cached_trainer(model, H, W, len(r.resblocks), x.device.index or 0, x.shape[0])
    single, layers = _net_parameters(model)
    running = _NetRunning(model)
    torch._foreach_add_(running.counters, 1)   # BatchNorm2d.forward: num_batches_tracked += 1
    geo = (16, model.policy_head.fc.out_features, model.value_head.fc0.out_features)
    return _NetFn.apply(tr, running, geo, x, *single, *layers)

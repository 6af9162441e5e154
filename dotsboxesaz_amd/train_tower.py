"""Training-mode forward/backward of ResNetZero's residual blocks on the hand-written HIP kernels of csrc/train.hip
(C ABI: dbaz_trainer_* in include/dbaz.h), exposed to torch autograd.

Reference: the `for block in self.resblocks` loop of ResNet.forward (nn.py:23-28) with ResBlock.forward (nn.py:48-57)
under `model.train(True)` in NeuralNetWrapper.train (nn.py:203-221).  torch is plumbing here: it owns the parameter and
gradient tensors and calls the C ABI with their device pointers on its current stream; the arithmetic of the blocks --
conv3x3, batch statistics, normalisation, ReLU, skip connections and all their gradients -- runs in train.hip.
"""
import ctypes as C

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
        self.rows, self.cols, self.channels, self.blocks, self.max_batch, self.device = rows, cols, channels, blocks, max_batch, device

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
        out = torch.empty_like(x)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self._ck(self._L.dbaz_trainer_forward(self.h, x.shape[0], x.data_ptr(), _ptr_array(conv_w), _ptr_array(conv_b),
                                              _ptr_array(bn_w), _ptr_array(bn_b), _ptr_array(run_mean), _ptr_array(run_var),
                                              out.data_ptr(), C.c_void_p(stream)))
        return out

    def backward(self, grad_out, bn_w, g_conv_w, g_conv_b, g_bn_w, g_bn_b):
        gx = torch.empty_like(grad_out)
        stream = torch.cuda.current_stream(grad_out.device).cuda_stream
        self._ck(self._L.dbaz_trainer_backward(self.h, grad_out.data_ptr(), _ptr_array(bn_w), gx.data_ptr(), _ptr_array(g_conv_w),
                                               _ptr_array(g_conv_b), _ptr_array(g_bn_w), _ptr_array(g_bn_b), C.c_void_p(stream)))
        return gx


class _TowerFn(torch.autograd.Function):
    """autograd node around the two C calls.  args: trainer, running stats (lists), x, then 4 tensors per layer
    (conv.weight, conv.bias, bn.weight, bn.bias) in layer order."""

    @staticmethod
    def forward(ctx, trainer, run_mean, run_var, x, *params):
        L = len(params) // 4
        cw, cb, bw, bb = ([params[4 * l + k].detach().contiguous() for l in range(L)] for k in range(4))
        xc = x.detach().contiguous().float()
        out = trainer.forward(xc, cw, cb, bw, bb, run_mean, run_var)
        ctx.trainer, ctx.bw, ctx.shapes = trainer, bw, [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = len(ctx.shapes) // 4
        dev = grad_out.device
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.shapes]
        gx = ctx.trainer.backward(grad_out.contiguous().float(), ctx.bw, [grads[4 * l] for l in range(L)], [grads[4 * l + 1] for l in range(L)],
                                  [grads[4 * l + 2] for l in range(L)], [grads[4 * l + 3] for l in range(L)])
        return (None, None, None, gx) + tuple(grads)


_trainers = {}


def supported(model, x):
    """The HIP tower handles ResNetZero containers with 64 channels on a CUDA/HIP tensor."""
    r = getattr(model, "resnet", None)
    return bool(x.is_cuda and r is not None and len(r.resblocks) > 0 and r.conv0.out_channels == 64)


def resblocks_forward(model, x):
    """x -> the output of model.resnet.resblocks in training mode (batch statistics; running stats and
    num_batches_tracked updated), differentiable."""
    r = model.resnet
    blocks = list(r.resblocks)
    H, W = x.shape[2], x.shape[3]
    key = (H, W, len(blocks), x.device.index or 0)
    tr = _trainers.get(key)
    if tr is None or tr.max_batch < x.shape[0]:
        if tr is not None:
            tr.close()
        tr = TowerTrainer(H - 1, W - 1, 64, len(blocks), max(int(x.shape[0]), 1), x.device.index or 0)
        _trainers[key] = tr
    params, rm, rv, nbt = [], [], [], []
    for b in blocks:
        for conv, bn in ((b.conv1, b.bn1), (b.conv2, b.bn2)):
            params += [conv.weight, conv.bias, bn.weight, bn.bias]
            rm.append(bn.running_mean)
            rv.append(bn.running_var)
            nbt.append(bn.num_batches_tracked)
    torch._foreach_add_(nbt, 1)   # one launch for the 2*blocks counters (BatchNorm2d.forward: num_batches_tracked += 1)
    return _TowerFn.apply(tr, rm, rv, x, *params)

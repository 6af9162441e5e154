"""Host-side mirror of the reference's network interface (nn.py, dots_boxes_nn.py).

`ResNetZero(params)` / `SimpleNN(params)` keep the reference's constructor, module tree
and state_dict key names (so `torch.load(fn)['model_dict']` checkpoints load unchanged,
nn.py:124-129), but they are WEIGHT CONTAINERS: the forward pass of the product runs in
the HIP engine (`NeuralNetWrapper.predict_sync` -> dbaz_nn_predict).  Calling
`forward()` on a container raises -- there is no torch/CPU fallback on the product path.
"""
import numpy as np
import torch
from torch import nn


def _get(d, k):
    return d[k] if isinstance(d, dict) else getattr(d, k)


class _ResBlock(nn.Module):
    def __init__(self, ch, k, n_groups=1):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, k, padding=(k - 1) // 2, groups=n_groups)
        self.bn1 = nn.BatchNorm2d(ch)
        self.conv2 = nn.Conv2d(ch, ch, k, padding=(k - 1) // 2, groups=n_groups)
        self.bn2 = nn.BatchNorm2d(ch)


class _ResNet(nn.Module):
    def __init__(self, in_channels, nb_channels, kernel_size, nb_blocks, n_groups=1, inner_channels=None,
                 pad_layer0=True):
        super().__init__()
        if kernel_size != 3 or inner_channels or not pad_layer0 or n_groups < 1 or nb_channels % n_groups:
            raise NotImplementedError("the HIP tower implements 3x3 kernels with a padded first layer and no inner conv "
                                      "(configuration.py:134-142); n_groups must divide nb_channels")
        # n_groups > 1 (nn.py:33-47,61-71): the grouped weights [ch][ch / groups][3][3] are expanded to the dense block-diagonal
        # form when they are loaded into the engine (Engine.load_state_dict) -- same sums, the off-diagonal products are exact zeros
        self.n_groups = n_groups
        self.conv0 = nn.Conv2d(in_channels, nb_channels, 3, padding=1)
        self.bn0 = nn.BatchNorm2d(nb_channels)
        self.resblocks = nn.Sequential(*[_ResBlock(nb_channels, kernel_size, n_groups) for _ in range(nb_blocks)])


class _PolicyHead(nn.Module):
    def __init__(self, in_channels, inner_channels, fc_in, nb_actions):
        super().__init__()
        self.conv0 = nn.Conv2d(in_channels, inner_channels, kernel_size=(1, 1))
        self.bn0 = nn.BatchNorm2d(inner_channels)
        self.fc = nn.Linear(fc_in, nb_actions)


class _ValueHead(nn.Module):
    def __init__(self, in_channels, inner_channels, fc_in, fc_inner):
        super().__init__()
        self.conv0 = nn.Conv2d(in_channels, inner_channels, kernel_size=(1, 1))
        self.bn0 = nn.BatchNorm2d(inner_channels)
        self.fc0 = nn.Linear(fc_in, fc_inner)
        self.fc1 = nn.Linear(fc_inner, 1)


class ResNetZero(nn.Module):
    """Reference: nn.py:108-129.  params.nn.model_parameters.{resnet,value_head,policy_head}."""

    kind = "resnet"

    def __init__(self, params):
        super().__init__()
        self.params = params
        mp = _get(_get(params, "nn"), "model_parameters")
        rp, vp, pp = dict(_get(mp, "resnet")), dict(_get(mp, "value_head")), dict(_get(mp, "policy_head"))
        self.bn_input = nn.BatchNorm2d(rp["in_channels"])
        self.resnet = _ResNet(**rp)
        self.value_head = _ValueHead(**vp)
        self.policy_head = _PolicyHead(**pp)
        self.shape = dict(channels=rp["nb_channels"], blocks=rp["nb_blocks"], head_channels=pp["inner_channels"],
                          value_fc=vp["fc_inner"])
        if vp["inner_channels"] != pp["inner_channels"]:
            raise NotImplementedError("policy and value heads must use the same inner_channels")

    def forward(self, x):
        raise RuntimeError("ResNetZero is a weight container on this path: evaluate through "
                           "NeuralNetWrapper.predict_sync (HIP engine); there is no torch fallback")

    def load_parameters(self, generation, to_device=None):
        fn = _get(_get(self.params, "nn"), "chkpts_filename").format(generation)
        self.load_state_dict(torch.load(fn, map_location="cpu", weights_only=True)["model_dict"])


class SimpleNN(nn.Module):
    """Reference: dots_boxes/dots_boxes_nn.py:61-105 (3x3 boards only).  Weight container."""

    kind = "simplenn"
    shape = {}

    def __init__(self, params=None, n_ch=256):
        super().__init__()
        self.params = params
        self.conv0 = nn.Conv2d(3, n_ch, 3, padding=1)
        self.bn0 = nn.BatchNorm2d(n_ch)
        for i in (1, 2, 3):
            setattr(self, "conv%d" % i, nn.Conv2d(n_ch, n_ch, 3, padding=1))
            setattr(self, "bn%d" % i, nn.BatchNorm2d(n_ch))
        self.conv4 = nn.Conv2d(n_ch, n_ch, 3, padding=0)
        self.bn4 = nn.BatchNorm2d(n_ch)
        self.fc0 = nn.Linear(1024, 512)
        self.bn_fc0 = nn.BatchNorm1d(512)
        self.fc1 = nn.Linear(512, 256)
        self.bn_fc1 = nn.BatchNorm1d(256)
        self.value_fc = nn.Linear(256, 1)
        self.policy_fc = nn.Linear(256, 32)

    def forward(self, x):
        raise RuntimeError("SimpleNN is a weight container on this path: evaluate through "
                           "NeuralNetWrapper.predict_sync (HIP engine); there is no torch fallback")

    def load_parameters(self, generation, to_device=None):
        fn = _get(_get(self.params, "nn"), "chkpts_filename").format(generation)
        self.load_state_dict(torch.load(fn, map_location="cpu", weights_only=True)["model_dict"])


def resnet_params(rows, cols, channels=64, blocks=20, head_channels=16, value_fc=8, n_groups=1):
    """The shipped `resnet` configuration (configuration.py:134-156) resized to a rows x cols board."""
    H, W = rows + 1, cols + 1
    return {"nn": {"model_parameters": {
        "resnet": {"pad_layer0": True, "in_channels": 3, "nb_channels": channels, "inner_channels": None,
                   "kernel_size": 3, "nb_blocks": blocks, "n_groups": n_groups},
        "policy_head": {"in_channels": channels, "inner_channels": head_channels, "fc_in": head_channels * H * W,
                        "nb_actions": 2 * H * W},
        "value_head": {"in_channels": channels, "inner_channels": head_channels, "fc_in": head_channels * H * W,
                       "fc_inner": value_fc}}, "pytorch_device": "cuda:0", "chkpts_filename": "model_gen{}.pt"}}


class NeuralNetWrapper:
    """Reference: nn.py:145-173 -- predict_sync(X [n,3,H,W]) -> (softmax p [n,A], v [n,1]) numpy float32."""

    def __init__(self, model, params=None, engine=None, rows=None, cols=None, n_slots=4096, device=0, nn_precision=None):
        from .engine import Engine
        self.params = params
        self.engine = engine
        self._own = engine is None
        if engine is None:
            if rows is None:
                raise ValueError("rows/cols (board size) required when no engine is given")
            self.engine = Engine(rows, cols, n_slots, evaluator=model.kind if model is not None else "resnet",
                                 device=device, nn_precision=nn_precision)
        self.model = None
        if model is not None:
            self.set_model(model)

    def set_model(self, model):
        self.model = model
        self.engine.load_state_dict(model.state_dict(), model.kind, **model.shape)

    def predict_sync(self, X):
        return self.engine.predict(np.asarray(X, dtype=np.float32))

    async def predict(self, X):
        return self.predict_sync(X)

    async def predict_from_game(self, game_state):
        return await self.predict([game_state.get_features()])

    async def __call__(self, X):
        return await self.predict(X)

    def train(self, train_dataset, val_dataset, writer, generation, hip_validation=True):
        """nn.py:175-274.  The optimizer step runs in torch on ROCm (see train.py); batches come from
        HBM when the datasets are train_data.ReplayDataset objects.  The trained weights are pushed
        back into the HIP engine so that the next self-play generation uses them.
        hip_validation: the validation passes (nn.py:223-246, model.train(False)) run on the HIP inference engine -- the same
        eval-mode network the reference evaluates, log p = log(softmax) instead of log_softmax: |d log p| <= 1e-4 wherever
        p >= 1e-3, validation losses within 1e-4 of torch's eval forward (tests/test_hip_coach.py) -- and fall back to torch's
        forward for a batch the engine rejects (SimpleNN in f16x3 has no exact-f32 safety net).  False: torch's forward."""
        from . import train as T
        from . import _lib
        import torch

        def eval_forward(model):
            # validation passes on the HIP inference engine: the epoch's weights go in once, the batches through predict_sync
            self.engine.load_state_dict(model.state_dict(), model.kind, **model.shape)

            def fwd(boards):
                try:
                    p, v = self.engine.predict(boards.detach().cpu().numpy())
                except _lib.DbazError:  # e.g. an activation beyond f16's range without a safety net: torch evaluates this batch
                    return T.training_forward(model, boards)
                lp = torch.log(torch.from_numpy(p).clamp_min(1e-38)).to(boards.device)
                return lp, torch.from_numpy(v).to(boards.device)
            return fwd

        last = T.train(self.model, self.params, train_dataset, val_dataset, writer, generation,
                       device="cuda:%d" % self.engine.cfg.device, eval_forward=eval_forward if hip_validation else None)
        self.model.to("cpu")
        self.set_model(self.model)
        return last

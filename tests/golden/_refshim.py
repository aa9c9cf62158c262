"""Import the *reference* (read-only, /root/reference) in the build container.

Only ``make_golden.py`` uses this; it never runs on the GPU box (the reference does not travel).
Shims per SURVEY.md App. C: fake torchvision.ops bits, empty wandb, the oracle's C restatement of
``sort_vertices`` (the CUDA op cannot be built here), and a no-op ``Tensor.cuda``.
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference/nerf_rpn"


def install():
    sys.dont_write_bytecode = True
    if "model.nerf_rpn" in sys.modules:
        return
    repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from oracle import geometry

    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")
    sd = types.ModuleType("torchvision.ops.stochastic_depth")
    misc = types.ModuleType("torchvision.ops.misc")

    class StochasticDepth(nn.Module):
        def __init__(self, p, mode):
            super().__init__()
            self.p, self.mode = p, mode

        def forward(self, x):
            if not self.training or self.p == 0.0:
                return x
            keep = 1.0 - self.p
            shape = [x.shape[0]] + [1] * (x.ndim - 1) if self.mode == "row" else [1] * x.ndim
            noise = torch.empty(shape, dtype=x.dtype, device=x.device).bernoulli_(keep)
            if keep > 0:
                noise.div_(keep)
            return x * noise

    class Permute(nn.Module):
        def __init__(self, dims):
            super().__init__()
            self.dims = dims

        def forward(self, x):
            return torch.permute(x, self.dims)

    class MLP(nn.Sequential):
        def __init__(self, in_channels, hidden_channels, norm_layer=None, activation_layer=nn.ReLU,
                     inplace=None, bias=True, dropout=0.0):
            layers, d = [], in_channels
            for h in hidden_channels[:-1]:
                layers += [nn.Linear(d, h, bias=bias), activation_layer(), nn.Dropout(dropout)]
                d = h
            layers += [nn.Linear(d, hidden_channels[-1], bias=bias), nn.Dropout(dropout)]
            super().__init__(*layers)

    def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2, reduction="none"):
        p = torch.sigmoid(inputs)
        ce = torch.nn.functional.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
        pt = p * targets + (1 - p) * (1 - targets)
        loss = ce * ((1 - pt) ** gamma)
        if alpha >= 0:
            loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
        return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss

    sd.StochasticDepth, misc.MLP, misc.Permute = StochasticDepth, MLP, Permute
    ops.sigmoid_focal_loss, ops.stochastic_depth, ops.misc = sigmoid_focal_loss, sd, misc
    tv.ops = ops
    sys.modules.update({"torchvision": tv, "torchvision.ops": ops,
                        "torchvision.ops.stochastic_depth": sd, "torchvision.ops.misc": misc})
    sys.modules["wandb"] = types.ModuleType("wandb")
    sv = types.ModuleType("sort_vertices")
    sv.sort_vertices_forward = lambda v, m, n: geometry.sort_vertices(v, m, n)
    sys.modules["sort_vertices"] = sv
    torch.Tensor.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)

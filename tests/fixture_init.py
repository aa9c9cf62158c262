"""Deterministic, construction-order-independent parameter init shared by the golden generator
(reference side) and the tests (oracle / HIP side).  Data helper only -- no reference code."""
import math
import zlib

import torch


def seeded_state(module, salt=0, bias_jitter=0.05):
    """Overwrite every parameter/buffer of ``module`` from a generator seeded by its key name.

    Convs/linears get N(0, 1/sqrt(fan_in))-ish weights, BN gets weight~U(0.5,1.5), bias~N(0,.1),
    running_mean~N(0,.1), running_var~U(0.5,1.5); biases get N(0, bias_jitter) so scores are
    tie-free.
    """
    sd = module.state_dict()
    for name, t in sd.items():
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + salt) & 0x7FFFFFFF)
        if not t.is_floating_point():
            continue
        if name.endswith("running_var"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif name.endswith("running_mean"):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.ndim == 1 and name.endswith("weight"):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif t.ndim == 1:
            v = torch.randn(t.shape, generator=g) * bias_jitter
        else:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (1.0 / math.sqrt(fan_in))
        t.copy_(v)
    module.load_state_dict(sd)
    return module

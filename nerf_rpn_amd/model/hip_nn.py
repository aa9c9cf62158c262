"""Executes standard ``torch.nn`` container modules (Conv3d / BatchNorm3d / ReLU / MaxPool3d / Sequential) on the HIP
kernels.  The nn modules only hold parameters under the reference's state-dict names; the arithmetic is
``nerf_rpn_amd.ops`` on channels-last tensors, with conv+BN+ReLU and conv+ReLU patterns fused the way the kernels
expect (bias/ReLU in the conv epilogue, ReLU inside the BN apply)."""
import torch
from torch import nn

from .. import ops


def _pack_of(mod):
    pk = mod.__dict__.get("_nrpn_pack")
    if pk is None:
        pk = ops.PackedWeight()
        mod.__dict__["_nrpn_pack"] = pk
    return pk


def conv3d(mod, x, relu=False, out_f32=False, segs=None, chain=0, stats=None, affine=None):
    """nn.Conv3d (k 1|3, stride 1, same padding; or the 4-channel k7 stem) on a channels-last tensor.
    ``segs``: x is a ragged list [1, sum voxels, 1, 1, C] of grids with these (X, Y, Z) dims (see ``ragged_cat``).
    ``affine``: (scale, shift) f32 [Cout] of an eval-mode BatchNorm folded behind this conv (``bn_fold``): y = conv_nobias * scale + shift."""
    k = mod.kernel_size[0]
    if k == 7:
        if mod.in_channels != 4 or mod.padding[0] != 3:
            raise NotImplementedError("k7 conv is only implemented for the 4-channel stem")
        cache = mod.__dict__.setdefault("_nrpn_stem", {})
        if affine is not None:
            return ops.StemFn.apply(x, mod.weight, mod.bias, mod.stride[0], cache, affine, relu)
        y = ops.StemFn.apply(x, mod.weight, mod.bias, mod.stride[0], cache)
        if relu:
            raise NotImplementedError("stem is always followed by BatchNorm in this model family")
        return y
    if k == 1 and mod.stride[0] > 1 and mod.padding[0] == 0:
        x = ops.SubsampleFn.apply(x, mod.stride[0])      # strided 1x1x1 conv = subsample + 1x1x1 GEMM
    elif k not in (1, 3) or mod.stride[0] != 1 or mod.padding[0] != k // 2:
        raise NotImplementedError(f"Conv3d k={k} stride={mod.stride} padding={mod.padding} has no HIP kernel yet")
    if segs is not None and (k == 7 or mod.stride[0] != 1):
        raise NotImplementedError("ragged voxel lists are supported by the stride-1 k1 / k3 convolutions only")
    mode = (relu, chain, stats, affine) if affine is not None else ((relu, chain, stats) if stats is not None else ((relu, chain) if chain else relu))
    return ops.ConvFn.apply(x, _pack_of(mod), mod.out_channels, mode, out_f32, 1 if segs is None else (1, tuple(segs)), mod.weight, mod.bias)


def ragged_cat(feats):
    """Pyramid levels [N,X_l,Y_l,Z_l,C] -> one ragged voxel list [1, sum N*X*Y*Z, 1, 1, C] (level-major, scene-major inside a level)
    plus the per-segment grid dims; layers that share weights across levels then run as ONE launch (ops.ConvFn with segs)."""
    import torch
    n, c = feats[0].shape[0], feats[0].shape[-1]
    segs = [tuple(int(v) for v in f.shape[1:4]) for f in feats for _ in range(n)]
    x = torch.cat([f.reshape(-1, c) for f in feats], dim=0)
    return x.view(1, x.shape[0], 1, 1, c), segs


def ragged_split(x, feats):
    """Inverse view of ``ragged_cat`` for a tensor with any channel count: per-level [N,X_l,Y_l,Z_l,C'] views."""
    outs, off = [], 0
    for f in feats:
        cnt = f.shape[0] * f.shape[1] * f.shape[2] * f.shape[3]
        outs.append(x[0, off:off + cnt, 0, 0, :].view(f.shape[0], f.shape[1], f.shape[2], f.shape[3], x.shape[-1]))
        off += cnt
    return outs


class BNCounters:
    """``num_batches_tracked`` of every BatchNorm3d under ``root`` as 0-dim views of ONE int64 tensor, so a training forward bumps
    all of them with a single launch (``step()``) instead of one tiny add per layer.  The buffers keep their names, values and
    state-dict behaviour; the caller then runs the layers with ``counted=True`` so they skip their own increment."""

    def __init__(self, root):
        self.root, self.flat, self.mods = root, None, None

    def step(self):
        if self.mods is None:
            self.mods = [m for m in self.root.modules()
                         if isinstance(m, nn.BatchNorm3d) and m.track_running_stats and m.num_batches_tracked is not None]
        if not self.mods:
            return
        stale = self.flat is None or any(m.num_batches_tracked.data_ptr() != self.flat[i].data_ptr() for i, m in enumerate(self.mods))
        if stale:      # first use, or the module was moved / reloaded with fresh buffers
            self.flat = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in self.mods]).contiguous()
            for i, m in enumerate(self.mods):
                m.num_batches_tracked = self.flat[i]
        if all(m.training for m in self.mods):
            self.flat.add_(1)
        else:          # some BatchNorm layers are frozen (eval mode inside a training backbone): only the live ones count
            for m in self.mods:
                if m.training:
                    m.num_batches_tracked.add_(1)


def bn_counters(root):
    c = root.__dict__.get("_nrpn_bn_counters")
    if c is None:
        c = BNCounters(root)
        root.__dict__["_nrpn_bn_counters"] = c
    return c


import os as _os

FUSED_BN_STATS = _os.environ.get("NRPN_BN_FUSED_STATS", "1") != "0"      # A/B switch; default on
FOLD_EVAL_BN = _os.environ.get("NRPN_BN_FOLD", "1") != "0"               # A/B switch; default on


def bn_fold(conv, bn):
    """(scale, shift) f32 [C] of an eval-mode BatchNorm3d folded behind ``conv`` (reference conv -> BN -> ReLU, feature_extractor.py:345-358):
    scale = gamma / sqrt(running_var + eps), shift = (conv bias - running_mean) * scale + beta.  Cached on the parameter / buffer
    versions and the raw-pointer weight epoch, so an eval forward costs no extra launches after the first."""
    ts = [conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
    key = tuple((t.data_ptr(), t._version) if t is not None else None for t in ts) + (bn.eps, ops._weight_epoch, ops.BN_STATS_EPOCH[0])
    ent = bn.__dict__.get("_nrpn_fold")
    if ent is None or ent[0] != key:
        with torch.no_grad():
            var = bn.running_var.detach().float()
            gamma = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(var)
            beta = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(var)
            scale = gamma * torch.rsqrt(var + bn.eps)
            cb = conv.bias.detach().float() if conv.bias is not None else torch.zeros_like(var)
            shift = (cb - bn.running_mean.detach().float()) * scale + beta
        ent = (key, scale.contiguous(), shift.contiguous())
        bn.__dict__["_nrpn_fold"] = ent
    return ent[1], ent[2]


def _can_fold(conv, bn, x):
    """eval-mode BatchNorm behind a conv the kernels can carry it in (stride-1 k1 / k3 and the stem), and nothing needs a gradient."""
    if not FOLD_EVAL_BN or bn.training or bn.running_mean is None:
        return False
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        return False
    k = conv.kernel_size[0]
    return k == 7 or k == 1 or (k == 3 and conv.stride[0] == 1)


def conv_bn(conv, bn, x, relu, counted=False):
    """conv -> BatchNorm3d [-> ReLU]: one launch in eval mode (BatchNorm folded into the conv epilogue), conv (+ fused statistics) ->
    normalisation pass in training mode."""
    if _can_fold(conv, bn, x):
        return conv3d(conv, x, relu=relu, affine=bn_fold(conv, bn))
    holder = {} if (FUSED_BN_STATS and (bn.training or bn.running_mean is None) and conv.kernel_size[0] != 7) else None
    return batch_norm(bn, conv3d(conv, x, stats=holder), relu, counted, holder)


def batch_norm(mod, x, relu, counted=False, stats=None):
    """``counted``: the caller already bumped this module's num_batches_tracked through a BNCounters.step() of this forward.
    ``stats``: the holder handed to the conv that produced ``x`` (conv3d(..., stats=holder)); if that launch left partial statistics in it,
    only their finish runs here instead of a statistics pass over ``x``."""
    training = mod.training or mod.running_mean is None
    if mod.training and mod.track_running_stats and mod.num_batches_tracked is not None and not counted:
        mod.num_batches_tracked.add_(1)
    return ops.BatchNormFn.apply(x, mod.weight, mod.bias, mod.running_mean, mod.running_var, training,
                                 mod.momentum if mod.momentum is not None else 0.1, mod.eps, relu,
                                 stats.get("partials") if (stats and training) else None)


def max_pool(mod, x):
    k = mod.kernel_size if isinstance(mod.kernel_size, int) else mod.kernel_size[0]
    s = mod.stride if isinstance(mod.stride, int) else mod.stride[0]
    p = mod.padding if isinstance(mod.padding, int) else mod.padding[0]
    return ops.MaxPoolFn.apply(x, k, s, p, bool(mod.ceil_mode))


def run_modules(mods, x, counted=False):
    """Run a flat list of nn modules on a channels-last tensor, fusing conv -> [BN] -> [ReLU] runs.
    ``counted``: see batch_norm."""
    mods = list(mods)
    i = 0
    while i < len(mods):
        m = mods[i]
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        nxt2 = mods[i + 2] if i + 2 < len(mods) else None
        if isinstance(m, nn.Conv3d):
            if isinstance(nxt, nn.BatchNorm3d) and _can_fold(m, nxt, x):
                # eval mode: BatchNorm (+ ReLU) are the conv's epilogue -- no separate normalisation pass over the activation
                fuse = isinstance(nxt2, nn.ReLU)
                x = conv3d(m, x, relu=fuse, affine=bn_fold(m, nxt))
                i += 3 if fuse else 2
            elif isinstance(nxt, nn.BatchNorm3d):
                # training-mode statistics come out of the conv's epilogue where its kernel has them (nrpn_conv3d_fwd_stats)
                holder = {} if (FUSED_BN_STATS and (nxt.training or nxt.running_mean is None)) else None
                x = conv3d(m, x, stats=holder)
                fuse = isinstance(nxt2, nn.ReLU)
                x = batch_norm(nxt, x, fuse, counted, holder)
                i += 3 if fuse else 2
            elif isinstance(nxt, nn.ReLU):
                x = conv3d(m, x, relu=True)
                i += 2
            else:
                x = conv3d(m, x)
                i += 1
        elif isinstance(m, nn.MaxPool3d):
            x = max_pool(m, x)
            i += 1
        elif isinstance(m, nn.Sequential):
            x = run_modules(m, x, counted)
            i += 1
        elif isinstance(m, nn.BatchNorm3d):
            fuse = isinstance(nxt, nn.ReLU)
            x = batch_norm(m, x, fuse, counted)
            i += 2 if fuse else 1
        else:
            raise NotImplementedError(f"no HIP kernel mapping for module {type(m).__name__}")
    return x


def conv_out(size, k, s, p):
    return tuple((int(g) + 2 * p - k) // s + 1 for g in size)


def module_out_size(mods, size):
    """Spatial size after a flat list of Conv3d / MaxPool3d / Sequential / pointwise modules (shape arithmetic only)."""
    for m in mods:
        if isinstance(m, nn.Sequential):
            size = module_out_size(list(m), size)
        elif isinstance(m, nn.Conv3d):
            size = conv_out(size, m.kernel_size[0], m.stride[0], m.padding[0])
        elif isinstance(m, nn.MaxPool3d):
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            st = m.stride if isinstance(m.stride, int) else m.stride[0]
            pd = m.padding if isinstance(m.padding, int) else m.padding[0]
            size = tuple(ops.query("pool_out_size", int(g), k, st, pd, int(bool(m.ceil_mode))) for g in size)
    return size


def as_ncdhw(x):
    """Channels-last [N,X,Y,Z,C] -> the reference's logical [N,C,X,Y,Z] (a free view)."""
    return x.permute(0, 4, 1, 2, 3)


def as_ndhwc(x, dtype):
    """Accept either a channels-last-backed NCDHW view (free) or a plain NCDHW tensor (converted by a kernel)."""
    cl = x.permute(0, 2, 3, 4, 1)
    if cl.is_contiguous() and cl.dtype == dtype:
        return cl
    return ops.to_channels_last(x, dtype)

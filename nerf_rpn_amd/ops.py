"""Tensor-level wrappers over the C ABI (``include/nerfrpn.h``).

PyTorch is used only for device memory, streams and autograd graph stitching: every function here hands raw device
pointers + the current HIP stream to ``libnerfrpn_hip.so``.  Activations are channels-last tensors of shape
``[N, X, Y, Z, C]`` (fp32 or bf16).  There is no CPU / eager fallback: non-CUDA tensors raise.
"""
import itertools
import math

import torch

from . import lib
from .lib import BF16, CONV_BIAS, CONV_OUT_F32, CONV_RELU, F32, call

# Size / plan queries of the C ABI are pure functions of (arguments, process-wide tool knobs): a training step asks the same ~40 of them
# ~600 times, each a ctypes crossing with argument conversion (~3 us).  Memoised per (name, arguments) and dropped whenever a
# nrpn_set_* knob is touched (lib.KNOB_EPOCH).  Queries that take an nrpn_conv_opts pointer are keyed on the descriptor's plan fields.
_QCACHE = {"epoch": -1, "map": {}}


def query(name, *args):
    if _QCACHE["epoch"] != lib.KNOB_EPOCH[0]:
        _QCACHE["epoch"], _QCACHE["map"] = lib.KNOB_EPOCH[0], {}
    m = _QCACHE["map"]
    key = (name, args)
    v = m.get(key)
    if v is None:
        v = m[key] = lib.query(name, *args)
    return v


def _query_opts(name, opts, *args):
    """``query`` for the *_ex entry points: the nrpn_conv_opts descriptor is the last argument; its plan fields (not its address) key the cache."""
    if _QCACHE["epoch"] != lib.KNOB_EPOCH[0]:
        _QCACHE["epoch"], _QCACHE["map"] = lib.KNOB_EPOCH[0], {}
    m = _QCACHE["map"]
    key = (name, args, opts.tile, opts.lds_dma, opts.kstep_bytes, opts.stagger, opts.big_split, opts.debug, bool(opts.scale), bool(opts.relu_mask))
    v = m.get(key)
    if v is None:
        v = m[key] = lib.query(name, *args, opts.ptr())
    return v


import os as _os
_SEPARATE_BIAS = bool(int(_os.environ.get('NRPN_SEPARATE_BIAS', '0')))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _s():
    """Raw handle of torch's current HIP stream on the current device (the stream every kernel of this module is enqueued on)."""
    if _raw_stream is not None:      # ~0.3 us; torch.cuda.current_stream() spends ~10 us in device-index / availability checks per call
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


# Weight-gradient side stream.  The wgrad of a conv layer (and the reduction of its slice partials into the gradient arena) is not on
# the backward's critical path -- only the optimiser needs it -- so ConvFn.backward can enqueue it on a second HIP stream: the
# memory-bound tails (slice reduction, BatchNorm backward of the next layer, split-K epilogues) then overlap with MFMA-bound kernels
# instead of each draining the chip on its own.  Consumers of the arena (optimiser step, bucket all-reduce) call wgrad_stream_join().
# On by default for arena training (every gradient of the layer has a GradSink); NRPN_WGRAD_STREAM=0 or set_wgrad_stream(False) keeps
# everything on the current stream.  Measured on the 160^3 VGG19-FPN step: 12.8 -> 12.0 ms.
_WGRAD_SIDE = {"enabled": _os.environ.get("NRPN_WGRAD_STREAM", "1") != "0", "streams": {}, "dirty": False, "prefetch": False, "keep": []}
# "keep": the tensors side-stream kernels of this backward pass read (dY, X), referenced until the main stream has joined the side stream.
# record_stream only defers the REUSE of their memory; it does not stop autograd from WRITING into them: the engine accumulates gradients
# in place when it holds the only reference (a residual join hands the same dY to both branches -- ScaleAddFn -- and the copy waiting in the
# other branch's input buffer is then added into on the main stream while the weight-gradient kernel still reads it: Swin, found in round 4
# by the graph-capture bit-identity tests).  A second reference makes the engine add out of place.


def set_wgrad_stream(enabled):
    _WGRAD_SIDE["enabled"] = bool(enabled)


def _wgrad_side_stream(device):
    if not _WGRAD_SIDE["enabled"] or device.type != "cuda":
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _WGRAD_SIDE["streams"].get(key)
    if st is None:
        # NRPN_WGRAD_PRIORITY: HIP stream priority of the weight-gradient stream (0 = default; positive = lower than the main stream's,
        # so the critical chain of the backward gets free workgroup slots first -- measured)
        prio = int(_os.environ.get("NRPN_WGRAD_PRIORITY", _WGRAD_SIDE.get("priority", 0)))
        st = _WGRAD_SIDE["streams"][key] = torch.cuda.Stream(device=device, priority=prio) if prio else torch.cuda.Stream(device=device)
    return st


def wgrad_stream_join():
    """Make the current stream wait for every weight-gradient kernel enqueued on the side stream so far."""
    if _WGRAD_SIDE["dirty"] or _WGRAD_SIDE["prefetch"]:      # wgrads of this backward pass / an operand refresh enqueued after the last step
        joined = True
        for st in _WGRAD_SIDE["streams"].values():
            cur = torch.cuda.current_stream(st.device)
            if cur == st:            # called from inside the side-stream context (a bucket completed by a wgrad): already ordered, but
                joined = False       # the main stream has not waited yet -- keep the flag for the next join
            else:
                cur.wait_stream(st)
        if joined:
            _WGRAD_SIDE["dirty"] = _WGRAD_SIDE["prefetch"] = False
            _WGRAD_SIDE["keep"] = []        # every later write of the main stream is ordered behind the side stream's reads now


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise lib.NrpnError("HIP op called with a non-CUDA tensor (the product path has no CPU fallback)")
        if not t.is_contiguous():
            raise lib.NrpnError(f"HIP op needs contiguous tensors, got strides {t.stride()} for shape {tuple(t.shape)}")


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise lib.NrpnError(f"unsupported dtype {t.dtype} (fp32 / bf16 only)")


def _f32(t):
    return t if t.dtype == torch.float32 else t.float()


# ======================================================================================================================
# rotated IoU / NMS / top-k
# ======================================================================================================================
def sort_vertices(vertices, mask, num_valid):
    """Drop-in for reference ``sort_vertices.sort_vertices_forward`` (cuda_op/sort_vert.cpp:6-33)."""
    vertices = vertices.float().contiguous()
    m8 = mask.to(torch.uint8).contiguous()
    nv = num_valid.to(torch.int32).contiguous()
    _chk(vertices, m8, nv)
    B, N, M = m8.shape
    out = torch.zeros((B, N, 9), dtype=torch.int32, device=vertices.device)
    call("sort_vertices_f32", _p(vertices), _p(m8), _p(nv), _p(out), B * N, M, _s())
    return out


def iou3d_pair(b1, b2):
    """Paired rotated 3D IoU, boxes [..., 7] -> [...] (reference cal_iou_3d)."""
    shape = b1.shape[:-1]
    a, b = _f32(b1).reshape(-1, 7).contiguous(), _f32(b2).reshape(-1, 7).contiguous()
    _chk(a, b)
    out = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    call("iou3d_obb_pair_f32", _p(a), _p(b), _p(out), a.shape[0], _s())
    return out.reshape(shape)


_IOU_LOSS_MODES = {"iou": 0, "linear_iou": 1, "giou": 2, "diou": 3}


class RotatedIoULossFn(torch.autograd.Function):
    """Per-pair IoU-type regression loss of RotatedIOULoss (reference model/rpn.py:133-164, fcos/loss.py:137-173): pred, target
    [n,7] -> (loss [n], iou [n]).  Forward and the gradient w.r.t. ``pred`` come out of one kernel (dual numbers over the reference's
    arithmetic, csrc/geomloss.hip); ``target`` is a constant."""

    @staticmethod
    def forward(ctx, pred, target, mode):
        p, t = _f32(pred).reshape(-1, 7).contiguous(), _f32(target).detach().reshape(-1, 7).contiguous()
        _chk(p, t)
        n = p.shape[0]
        loss = torch.empty(n, dtype=torch.float32, device=p.device)
        iou = torch.empty(n, dtype=torch.float32, device=p.device)
        grad = torch.empty((n, 7), dtype=torch.float32, device=p.device)
        call("rotated_iou_loss_f32", _p(p), _p(t), n, _IOU_LOSS_MODES[mode], _p(loss), _p(grad), _p(iou), _s())
        ctx.save_for_backward(grad)
        ctx.shape = pred.shape
        ctx.mark_non_differentiable(iou)
        return loss.reshape(pred.shape[:-1]), iou.reshape(pred.shape[:-1])

    @staticmethod
    def backward(ctx, g_loss, _g_iou):
        (grad,) = ctx.saved_tensors
        return (grad * g_loss.reshape(-1, 1)).reshape(ctx.shape), None, None


def rotated_iou_loss(pred, target, mode):
    """-> (per-pair loss, per-pair IoU); differentiable in ``pred``."""
    return RotatedIoULossFn.apply(pred, target, mode)


_SAMPLE_WS = {}


def sample_pos_neg(labels, batch, max_pos, seed, extra_flags=None, before_readback=None):
    """Balanced sampler of every scene in ONE host read-back (reference BalancedPositiveNegativeSampler, model/utils.py:35-98).
    labels: list of [T] float32 device tensors; seed: python int (one draw per call; scene i uses seed + i).
    -> ([(pos_i, neg_i)], host list of ``extra_flags``): int64 ascending index tensors; ``extra_flags`` (a list of 0-dim device tensors)
    ride on the same device->host copy so that callers with their own pending checks do not synchronise a second time.
    ``before_readback(out_pos, out_neg, counts)``: optional; enqueues more work on the sampler's raw output (device int64 [n, max_pos],
    int64 [n, batch], int32 (kp, kn, err) per scene) and returns ``(int32 device vector, finish)``: the vector rides on the same copy and
    ``finish(host values)`` is called after it (the cone lists of the RPN head are built this way, ``cone_build``)."""
    n = len(labels)
    dev = labels[0].device
    for lab in labels:
        if lab.dtype != torch.float32 or lab.dim() != 1:
            raise TypeError("sample_pos_neg: labels must be 1-D float32")
        _chk(lab.contiguous())
    wsb = int(query("sample_workspace_bytes"))
    key = (str(dev), n, batch, max_pos, _s())
    if key not in _SAMPLE_WS:          # per-scene scratch of this stream, reused every step (stream-ordered)
        _SAMPLE_WS[key] = torch.empty((n, (wsb + 7) // 8), dtype=torch.int64, device=dev)
    ws = _SAMPLE_WS[key]
    out_pos = torch.empty((n, max(max_pos, 1)), dtype=torch.int64, device=dev)
    out_neg = torch.empty((n, max(batch, 1)), dtype=torch.int64, device=dev)
    nx = len(extra_flags) if extra_flags else 0
    counts = torch.empty(n * 3 + nx, dtype=torch.int32, device=dev)
    for i, lab in enumerate(labels):
        call("sample_pos_neg", _p(lab.contiguous()), lab.numel(), int(max_pos), int(batch), (int(seed) + i) & 0x7FFFFFFFFFFFFFFF, _p(ws[i]),
             _p(out_pos[i]), _p(out_neg[i]), _p(counts[3 * i:]), _s())
    if nx:
        counts[3 * n:] = torch.stack([f.reshape(()) for f in extra_flags]).to(torch.int32)
    finish = None
    if before_readback is not None:
        more, finish = before_readback(out_pos, out_neg, counts)
        host = torch.cat([counts, more.to(torch.int32).reshape(-1)]).cpu().tolist()      # the one synchronisation
        finish(host[3 * n + nx:])
        host = host[:3 * n + nx]
    else:
        host = counts.cpu().tolist()                  # the one synchronisation
    res = []
    for i in range(n):
        kp, kn, err = host[3 * i:3 * i + 3]
        if err:
            raise RuntimeError("sample_pos_neg: a key bin holds more candidates than the kernel's LDS sort (see nrpn_sample_pos_neg)")
        res.append((out_pos[i, :kp], out_neg[i, :kn]))
    return res, host[3 * n:]


def projection_loss(pred, target, views, intrinsics, beta, max_mesh_dim):
    """2-D projection smooth-L1 of the RPN (value only, no gradient): pred / target [n,6|7], views [4,4,4], intrinsics [3,3] -> scalar
    (reference rpn.py:37-102, 421-453)."""
    if pred.shape != target.shape or pred.shape[1] not in (6, 7):
        raise ValueError(f"projection_loss: pred {tuple(pred.shape)} and target {tuple(target.shape)} must both be [n,6] or [n,7]")
    pred, target = _f32(pred.detach()).contiguous(), _f32(target.detach()).contiguous()
    views, intrinsics = views.contiguous(), intrinsics.contiguous()
    _chk(pred, target, views, intrinsics)
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    call("projection_loss_f32", _p(pred), _p(target), pred.shape[0], pred.shape[1], _p(views), _p(intrinsics), float(beta), float(max_mesh_dim),
         _p(out), _s())
    return out[0]


def iou3d_matrix(a, b):
    """All-pairs IoU [n,w] x [m,w] -> [n,m], w = 6 (AABB) or 7 (OBB) (reference box_iou_3d)."""
    if a.shape[1] != b.shape[1] or a.shape[1] not in (6, 7):
        raise ValueError(f"The second dimension of boxes1 and boxes2 should be the same, both 6 or 7. But get {a.shape[1]} and {b.shape[1]}.")
    a, b = _f32(a).contiguous(), _f32(b).contiguous()
    _chk(a, b)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    for i in range(0, a.shape[0], 32768):
        part = a[i:i + 32768]
        call("iou3d_matrix_f32", _p(part), _p(b), _p(out[i:]), part.shape[0], b.shape[0], a.shape[1], _s())
    return out


def nms3d_sorted(boxes, levels, thr, count=None):
    """Greedy NMS on boxes already score-descending inside each (non-decreasing) level run -> uint8 keep mask."""
    boxes = _f32(boxes).contiguous()
    _chk(boxes, levels, count)
    n = boxes.shape[0]
    keep = torch.empty(n, dtype=torch.uint8, device=boxes.device)
    if n == 0:
        return keep
    ws = torch.empty(query("nms3d_workspace_bytes", n), dtype=torch.uint8, device=boxes.device)
    call("nms3d", _p(boxes), _p(levels), _p(count), n, boxes.shape[1], float(thr), _p(keep), _p(ws), _s())
    return keep


def segmented_topk(scores, offsets, k):
    """Per-segment top-k in (score desc, index asc) order.  offsets: python list of nseg+1 ints."""
    import ctypes
    scores = scores.contiguous()
    _chk(scores)
    nseg = len(offsets) - 1
    idx = torch.empty((nseg, k), dtype=torch.int32, device=scores.device)
    val = torch.empty((nseg, k), dtype=torch.float32, device=scores.device)
    host = (ctypes.c_int64 * (nseg + 1))(*[int(o) for o in offsets])
    nbytes = int(query("segmented_topk_workspace_bytes", nseg, int(k)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=scores.device)
    call("segmented_topk_f32_ws", _p(scores), ctypes.addressof(host), nseg, int(k), _p(idx), _p(val), _p(ws), nbytes, _s())
    return idx, val


def argsort_desc(scores):
    """(score desc, index asc) permutation of a 1-D tensor with n <= 16384 (deterministic argsort)."""
    n = scores.numel()
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=scores.device)
    if n > 16384:
        raise lib.NrpnError("argsort_desc supports at most 16384 elements")
    idx, _ = segmented_topk(_f32(scores).reshape(-1), [0, n], n)
    return idx[0].long()


# ======================================================================================================================
# anchors / coders
# ======================================================================================================================
ANCHOR_SIZES = ((8,), (16,), (32,), (64,))
ASPECT_RATIOS = (((1., 1., 1.), (1., 1., 2.), (1., 2., 2.), (1., 1., 3.), (1., 3., 3.)),) * 4


def unique_ratio_permutations(ratios):
    """The reference iterates ``set(itertools.permutations(r))`` per ratio (anchor.py:57-60); CPython's hash order of
    float tuples is deterministic, so the same expression reproduces the channel <-> anchor mapping of checkpoints."""
    out = []
    for r in ratios:
        out += list(set(itertools.permutations(r)))
    return out


def base_anchor_table(scales, ratios):
    """anchor.py:49-82 (is_normalized=False): [A,6] rounded half-to-even, on the host."""
    r = torch.tensor(unique_ratio_permutations(ratios), dtype=torch.float32)
    s = torch.as_tensor(scales, dtype=torch.float32)
    e = (r[:, None, :] * s[None, :, None]).reshape(-1, 3)
    return (torch.cat([-e, e], dim=1) / 2).round()


class AnchorTable:
    """Device-side description of the anchor pyramid (layout in nerfrpn.h); anchors are computed from flat indices."""

    def __init__(self, mesh_size, grids, sizes=ANCHOR_SIZES, ratios=ASPECT_RATIOS, device="cuda"):
        bases = [base_anchor_table(s, r) for s, r in zip(sizes, ratios)]
        self.A = bases[0].shape[0]
        if any(b.shape[0] != self.A for b in bases):
            raise lib.NrpnError("all pyramid levels must have the same number of anchors per cell")
        self.grids = [tuple(int(v) for v in g) for g in grids]
        self.strides = [tuple(int(mesh_size[i]) // g[i] for i in range(3)) for g in self.grids]
        self.cells = [g[0] * g[1] * g[2] for g in self.grids]
        self.counts = [c * self.A for c in self.cells]
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        self.total = self.offsets[-1]
        L = len(self.grids)
        words = torch.zeros(int(query("anchor_table_words", L, self.A)), dtype=torch.int32)
        words[0], words[1] = L, self.A
        for l, (g, s) in enumerate(zip(self.grids, self.strides)):
            first = self.offsets[l]
            lo = first & 0xFFFFFFFF
            lo = lo - (1 << 32) if lo >= (1 << 31) else lo
            words[2 + 8 * l: 2 + 8 * l + 8] = torch.tensor([g[0], g[1], g[2], s[0], s[1], s[2], lo, first >> 32], dtype=torch.int32)
        fl = torch.cat([b.reshape(-1) for b in bases]).view(torch.int32)
        words[2 + 8 * L:] = fl
        self.words = words.to(device)
        self.base = bases

    def level_of(self, device):
        return torch.cat([torch.full((c,), i, dtype=torch.int64, device=device) for i, c in enumerate(self.counts)])


def anchors(table, sel=None, count=None):
    count = table.total if sel is None else sel.numel()
    out = torch.empty((count, 6), dtype=torch.float32, device=table.words.device)
    _chk(sel)
    call("anchors_f32", _p(table.words), _p(sel), count, _p(out), _s())
    return out


def decode_boxes(table, deltas, sel, coder):
    """deltas [T, 6|8] (flat anchor order), sel int64 [K] or None -> boxes [K, 6|7]."""
    deltas = _f32(deltas).contiguous()
    _chk(deltas, sel)
    count = deltas.shape[0] if sel is None else sel.numel()
    out = torch.empty((count, 7 if coder else 6), dtype=torch.float32, device=deltas.device)
    call("decode_boxes_f32", _p(table.words), _p(deltas), _p(sel), count, int(coder), _p(out), _s())
    return out


def encode_boxes(table, gt, sel, coder):
    gt = _f32(gt).contiguous()
    _chk(gt, sel)
    count = gt.shape[0]
    out = torch.empty((count, 8 if coder else 6), dtype=torch.float32, device=gt.device)
    call("encode_boxes_f32", _p(table.words), _p(gt), _p(sel), count, int(coder), _p(out), _s())
    return out


def coder_pairs(inp, anchor_boxes, coder, encode):
    inp, anchor_boxes = _f32(inp).contiguous(), _f32(anchor_boxes).contiguous()
    _chk(inp, anchor_boxes)
    if inp.shape[0] != anchor_boxes.shape[0]:
        raise AssertionError("coder: row count mismatch")
    width = (8 if coder else 6) if encode else (7 if coder else 6)
    out = torch.empty((inp.shape[0], width), dtype=torch.float32, device=inp.device)
    call("coder_pairs_f32", _p(inp), _p(anchor_boxes), inp.shape[0], int(coder), int(bool(encode)), _p(out), _s())
    return out


def obb_to_aabb(obb):
    obb = _f32(obb).contiguous()
    _chk(obb)
    out = torch.empty((obb.shape[0], 6), dtype=torch.float32, device=obb.device)
    call("obb_to_aabb_f32", _p(obb), _p(out), obb.shape[0], _s())
    return out


# ======================================================================================================================
# proposal filter / target assignment / losses
# ======================================================================================================================
def filter_candidates(boxes, logits, levels, valid, grid_size, min_size, score_thresh, fix_obb_clip=False):
    import ctypes
    _chk(boxes, logits, levels, valid)
    n, w = boxes.shape
    dev = boxes.device
    ob = torch.empty((n, w), dtype=torch.float32, device=dev)
    os_ = torch.empty(n, dtype=torch.float32, device=dev)
    ol = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(16, query("filter_workspace_bytes", n, w)), dtype=torch.uint8, device=dev)
    host = (ctypes.c_float * 3)(*[float(v) for v in grid_size])
    call("filter_candidates_f32", _p(boxes), _p(logits), _p(levels), _p(valid), n, w, ctypes.addressof(host), float(min_size),
         float(score_thresh), int(bool(fix_obb_clip)), _p(ob), _p(os_), _p(ol), _p(cnt), _p(ws), _s())
    return ob, os_, ol, cnt


def select_kept(boxes, scores, levels, keep, count, post_top_n):
    _chk(boxes, scores, levels, keep, count)
    n, w = boxes.shape
    dev = boxes.device
    ob = torch.empty((post_top_n, w), dtype=torch.float32, device=dev)
    os_ = torch.empty(post_top_n, dtype=torch.float32, device=dev)
    ol = torch.empty(post_top_n, dtype=torch.float32, device=dev)
    oc = torch.zeros(1, dtype=torch.int32, device=dev)
    call("select_kept_f32", _p(boxes), _p(scores), _p(levels), _p(keep), _p(count), n, w, int(post_top_n), _p(ob), _p(os_), _p(ol),
         _p(oc), _s())
    return ob, os_, ol, oc


def match_anchors(table, gt_aabb, fg, bg, ori_size=None):
    """labels f32 [T] in {1,0,-1} and clamped match index int32 [T] (reference assign_targets_to_anchors + Matcher)."""
    import ctypes
    gt_aabb = _f32(gt_aabb).contiguous()
    _chk(gt_aabb)
    dev = gt_aabb.device
    labels = torch.empty(table.total, dtype=torch.float32, device=dev)
    matched = torch.empty(table.total, dtype=torch.int32, device=dev)
    ws = torch.empty(gt_aabb.shape[0], dtype=torch.float32, device=dev)
    host = None
    hp = 0
    if ori_size is not None:
        host = (ctypes.c_float * 3)(*[float(v) for v in ori_size])
        hp = ctypes.addressof(host)
    call("match_anchors_f32", _p(table.words), table.total, _p(gt_aabb), gt_aabb.shape[0], float(fg), float(bg), hp, _p(labels),
         _p(matched), _p(ws), _s())
    return labels, matched


class SampledLossFn(torch.autograd.Function):
    """BCE-with-logits (mean over pos+neg) and smooth-L1 (sum over pos / (|pos|+|neg|)) with fused backward
    (reference compute_loss, rpn.py:372-419).  logits [M], deltas [M,dw] flat over the batch."""

    @staticmethod
    def forward(ctx, logits, deltas, targets, pos, neg, beta):
        logits, deltas = logits.contiguous(), deltas.contiguous()
        targets = targets.contiguous()
        _chk(logits, deltas, targets, pos, neg)
        out = torch.empty(2, dtype=torch.float32, device=logits.device)
        g_logits = torch.zeros_like(logits)
        g_deltas = torch.zeros_like(deltas)
        call("rpn_sampled_loss_f32", _p(logits), _p(deltas), deltas.shape[-1], _p(targets), _p(pos), pos.numel(), _p(neg), neg.numel(),
             float(beta), _p(out), _p(g_logits), _p(g_deltas), _s())
        ctx.save_for_backward(g_logits, g_deltas)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_obj, g_reg):
        g_logits, g_deltas = ctx.saved_tensors
        return g_logits * g_obj, g_deltas * g_reg, None, None, None, None


class FlattenHeadFn(torch.autograd.Function):
    """Per-level head rows [N, cells_l, ld] -> logits [N, T], deltas [N, T, dw] in the reference's (level, x, y, z, a) order
    (concat_box_prediction_layers, rpn.py:105-130); backward scatters dense gradients back into the padded head rows."""

    @staticmethod
    def forward(ctx, A, dw, grad_dtype, *heads):
        n = heads[0].shape[0]
        cells = [h.shape[1] for h in heads]
        T = sum(cells) * A
        dev = heads[0].device
        logits = torch.empty((n, T), dtype=torch.float32, device=dev)
        deltas = torch.empty((n, T, dw), dtype=torch.float32, device=dev)
        off = 0
        for h, c in zip(heads, cells):
            _chk(h)
            for i in range(n):
                call("head_flatten_f32", _p(h[i]), c, h.shape[2], A, dw, _p(logits[i, off:]), _p(deltas[i, off:]), _s())
            off += c * A
        ctx.meta = (A, dw, cells, [h.shape[2] for h in heads], grad_dtype)
        return logits, deltas

    @staticmethod
    def backward(ctx, g_logits, g_deltas):
        A, dw, cells, lds, grad_dtype = ctx.meta
        g_logits, g_deltas = g_logits.contiguous(), g_deltas.contiguous()
        n = g_logits.shape[0]
        outs = []
        off = 0
        for c, ld in zip(cells, lds):
            d = torch.empty((n, c, ld), dtype=grad_dtype, device=g_logits.device)
            for i in range(n):
                call("head_unflatten", _p(g_logits[i, off:]), _p(g_deltas[i, off:]), c, ld, A, dw, 0, _p(d[i]), _dt(d), _s())
            outs.append(d)
            off += c * A
        return (None, None, None, *outs)


# ======================================================================================================================
# conv / norm / pool with autograd
# ======================================================================================================================
class GradSink:
    """Direct gradient destination of a parameter inside a flat gradient arena (set by engine.FlatTrainer as
    ``param._nrpn_sink``).  Backward kernels accumulate straight into ``slot`` and call ``notify()`` instead of returning a
    gradient tensor for autograd to add -- this removes one elementwise kernel per parameter per use.
    ``flat``: for weights the trainer keeps in the forward GEMM layout [taps][Cout][Cin], the contiguous 1-D arena range of the
    gradient in THAT layout (the sum of a wgrad's slice partials goes there without any layout shuffle); None otherwise."""
    __slots__ = ("slot", "notify", "flat")

    def __init__(self, slot, notify, flat=None):
        self.slot, self.notify, self.flat = slot, notify, flat


class ArenaWeights:
    """GEMM operands of the weights a trainer keeps in the forward GEMM layout inside its flat fp32 arena (engine.FlatTrainer).

    forward operand  = the master arena itself (fp32) or its bf16 shadow -- the same element order, written by the AdamW kernel as a
                       second output, so there is no per-step repack;
    dgrad operand    = [taps reversed][Cin][Cout] copies of ALL weights, refreshed by ONE batched launch per optimiser step.
    ``epoch`` counts parameter updates (optimiser steps, or torch-side writes noticed through tensor._version)."""

    def __init__(self, master):
        self.master = master
        self.entries = []             # (offset, taps, cout, cin, param)
        self.versions = []
        self.epoch = 0
        self.shadow = None            # bf16 [len(master)]
        self.shadow_epoch = -1
        self.trans = {}               # dtype -> [tensor, epoch]
        self.table = None
        self.launches = {"cast": 0, "transpose": 0}

    def add(self, param, offset, taps, cout, cin):
        param._nrpn_arena = (self, len(self.entries))
        self.entries.append((int(offset), int(taps), int(cout), int(cin), param))
        self.versions.append(param._version)
        self.table = None

    def bump(self, shadow_written=False):
        """The master arena was rewritten (optimiser step)."""
        self.epoch += 1
        if shadow_written and self.shadow is not None:
            self.shadow_epoch = self.epoch

    def _check(self, idx):
        if self.entries[idx][4]._version != self.versions[idx]:      # torch-side write (load_state_dict, manual init): everything is stale
            self.versions = [e[4]._version for e in self.entries]
            self.epoch += 1

    def shadow_ptr(self):
        """bf16 shadow for the AdamW kernel to refresh (None until a bf16 forward asked for it)."""
        return self.shadow

    def fwd(self, idx, dtype):
        self._check(idx)
        off, taps, cout, cin, _ = self.entries[idx]
        if dtype == torch.float32:
            return self.master[off:off + taps * cout * cin].view(taps, cout, cin)
        if self.shadow is None:
            self.shadow = torch.empty(self.master.numel(), dtype=torch.bfloat16, device=self.master.device)
        if self.shadow_epoch != self.epoch:
            call("cast", _p(self.master), _p(self.shadow), self.master.numel(), F32, BF16, _s())
            self.launches["cast"] += 1
            self.shadow_epoch = self.epoch
        return self.shadow[off:off + taps * cout * cin].view(taps, cout, cin)

    def dgrad(self, idx, dtype):
        self._check(idx)
        off, taps, cout, cin, _ = self.entries[idx]
        ent = self.trans.get(dtype)
        if ent is None:
            ent = [torch.empty(self.master.numel(), dtype=dtype, device=self.master.device), -1]
            self.trans[dtype] = ent
        if ent[1] != self.epoch:
            self._transpose(dtype)
        return ent[0][off:off + taps * cout * cin].view(taps, cin, cout)

    def _transpose(self, dtype):
        ent = self.trans[dtype]
        if self.table is None:
            rows, prefix = [], [0]
            for (o, t, co, ci, _) in self.entries:
                rows.append([o, t, co, ci])
                prefix.append(prefix[-1] + t * ((co + 63) // 64) * ((ci + 63) // 64))
            dev = self.master.device
            self.table = (torch.tensor(rows, dtype=torch.int64, device=dev), torch.tensor(prefix, dtype=torch.int32, device=dev), prefix[-1])
        tab, prefix, total = self.table
        call("transpose_weights", _p(self.master), _p(ent[0]), _p(tab), _p(prefix), len(self.entries), total, F32 if dtype == torch.float32 else BF16, _s())
        self.launches["transpose"] += 1
        ent[1] = self.epoch

    def prefetch_dgrad(self):
        """Called by the trainer right after an optimiser step: refresh the dgrad operands on the weight-gradient side stream, off the
        next forward's critical path (they are first read by the next BACKWARD, which waits for the event recorded here)."""
        side = _wgrad_side_stream(self.master.device)
        if side is None or not self.trans or self.table is None:
            return
        side.wait_stream(torch.cuda.current_stream(self.master.device))      # behind the AdamW kernel that rewrote the master arena
        with torch.cuda.stream(side):
            for dtype in list(self.trans):
                if self.trans[dtype][1] != self.epoch:
                    self._transpose(dtype)
            ev = torch.cuda.Event()
            ev.record(side)
        _DGRAD_READY["event"] = ev
        _DGRAD_READY["waited"] = set()
        _WGRAD_SIDE["prefetch"] = True   # a step() with no backward in between still joins the side stream before it rewrites the arena
                                         # (a flag of its own: "dirty" also decides whether a backward pass queues its end-of-pass join)


_DGRAD_READY = {"event": None, "waited": set()}


def _wait_dgrad_operands():
    """First dgrad of a stream after an optimiser step: wait for the operand refresh enqueued on the side stream."""
    ev = _DGRAD_READY["event"]
    if ev is not None and torch.cuda.is_current_stream_capturing():
        return      # graph capture (graphs.GraphedBackbone): the replay is ordered behind the refresh by an eager wait on the launching stream
    if ev is not None:
        sid = _s()
        if sid not in _DGRAD_READY["waited"]:
            torch.cuda.current_stream().wait_event(ev)
            _DGRAD_READY["waited"].add(sid)


def _sink(t):
    return getattr(t, "_nrpn_sink", None) if t is not None else None


# Parameters can be rewritten behind autograd's back (engine.FlatTrainer updates its flat arena through the raw-pointer AdamW
# kernel, which never bumps tensor._version), so every GEMM-layout cache is also keyed on this epoch; whoever writes
# parameter memory outside torch calls ``weights_changed()``.
_weight_epoch = 0
PACK_COUNT = {"conv": 0, "stem": 0}      # pack launches so far (tests assert packs == modules x optimiser steps)


def a2a_reduce(recv, bucket, rank, world):
    """recv: bf16 [world * chunk] (chunk ``rank`` of every peer's bucket, peer-major), bucket: this rank's fp32 bucket [world * chunk] -> bf16
    [chunk] = sum over the ranks in ascending order in fp32, the own chunk read from the fp32 bucket (engine.FlatTrainer, a2a_bf16)."""
    chunk = bucket.numel() // world
    _chk(recv, bucket)
    out = torch.empty(chunk, dtype=torch.bfloat16, device=bucket.device)
    call("a2a_reduce_bf16", _p(recv), bucket.data_ptr() + rank * chunk * 4, rank, world, chunk, _p(out), _s())
    return out


def weights_changed():
    global _weight_epoch
    _weight_epoch += 1


class PackedWeight:
    """GEMM-layout copies of one or more reference-layout conv weights sharing a GEMM (rows_total rows), refreshed when
    the parameters change (tensor._version for torch-side writes, the module weight epoch for raw-pointer writes)."""

    def __init__(self):
        self.key = None
        self.fwd = None
        self.dgrad = None

    def get(self, weights, dtype, rows_total, need_dgrad, cin=None):
        cin = weights[0].shape[1] if cin is None else cin     # nn.Linear [out,in] and a flattened patch conv are taps == 1
        taps = weights[0][0].numel() // cin
        arena = getattr(weights[0], "_nrpn_arena", None) if len(weights) == 1 else None
        if arena is not None and rows_total == weights[0].shape[0] and arena[0].entries[arena[1]][1:4] == (taps, rows_total, cin):
            # master weights already live in the GEMM layout inside the trainer's arena: no packing at all
            aw, idx = arena
            return aw.fwd(idx, dtype), (aw.dgrad(idx, dtype) if need_dgrad else None)
        key = tuple((w.data_ptr(), w._version) for w in weights) + (dtype, rows_total, need_dgrad, _weight_epoch)
        if key == self.key:
            return self.fwd, self.dgrad
        PACK_COUNT["conv"] += 1
        dev = weights[0].device
        padded = rows_total != sum(w.shape[0] for w in weights)
        alloc = torch.zeros if padded else torch.empty
        fwd = alloc((taps, rows_total, cin), dtype=dtype, device=dev)
        dgrad = alloc((taps, cin, rows_total), dtype=dtype, device=dev) if need_dgrad else None
        row = 0
        for w in weights:
            wc = w.detach().contiguous()
            call("pack_conv_weight", _p(wc), w.shape[0], cin, taps, _dt(fwd), _p(fwd), _p(dgrad), rows_total, row, _s())
            row += w.shape[0]
        self.key, self.fwd, self.dgrad = key, fwd, dgrad
        return fwd, dgrad



# ======================================================================================================================
# bf16x3: the parity-grade fast mode (round 5).  fp32 activations / weights / gradients everywhere; the dense and row-list 3x3x3
# convolutions -- 95 % of the step's FLOPs -- multiply SPLIT operands (hi = bf16(x), lo = bf16(x - hi)) on the bf16 MFMA kernels:
# x * w ~ hi*whi + hi*wlo + lo*whi, fp32 accumulation, fp32 rows out (csrc/elementwise.hip: split_bf16x3_kernel).  The K axis is tripled,
# the kernels are the bf16 ones unchanged.  SPLIT3 decides HOW an fp32 convolution is computed, not what it returns (to fp32 accumulation
# error) -- in the spirit of torch.backends.cuda.matmul.allow_tf32.  A model raises it for the duration of its own forward pass
# (NeRFRegionProposalNetwork.set_compute_dtype("bf16x3"), run_rpn.py --dtype bf16x3; backward passes follow what their forward recorded);
# NRPN_BF16X3=1 turns it on for the whole process.
# ======================================================================================================================
SPLIT3 = [_os.environ.get("NRPN_BF16X3", "0") == "1"]
_ACT_I, _W_I = 0b100, 0b010          # interleaved [rows][3C]: activations (hi | hi | lo), weights (hi | lo | hi)
_X_P, _DY_P = 0b010, 0b100           # planes stacked on the batch axis: x (hi ; lo ; hi), dy (hi ; hi ; lo)


def split3(src, ipattern=None, nplanes=0, ppattern=0):
    """src f32 [..., C] -> (bf16 [..., 3C] interleaved or None, bf16 [nplanes, ..., C] planes or None); see nrpn_split_bf16x3."""
    src = src.contiguous()
    _chk(src)
    c = src.shape[-1]
    inter = torch.empty(tuple(src.shape[:-1]) + (3 * c,), dtype=torch.bfloat16, device=src.device) if ipattern is not None else None
    planes = torch.empty((nplanes,) + tuple(src.shape), dtype=torch.bfloat16, device=src.device) if nplanes else None
    call("split_bf16x3", _p(src), src.numel() // c, c, _p(inter), ipattern or 0, _p(planes), nplanes, ppattern, _s())
    return inter, planes


def _x3_ok(x, ksize, segs, rows_total, differentiable):
    # ragged voxel lists (the dense head over all pyramid levels) only when nothing is differentiated: their weight gradient has no batch
    # axis to stack on and the split dgrad launch knows no segments.  `differentiable` is decided from the operands -- inside
    # autograd.Function.forward grad mode is always off, so torch.is_grad_enabled() says nothing there
    return (SPLIT3[0] and x.dtype == torch.float32 and ksize == 3 and (segs is None or not differentiable)
            and x.shape[-1] % 32 == 0 and rows_total % 32 == 0)


def _x3_weights(pack, weights, wp, wpd):
    """Split forms of a conv's fp32 GEMM operands (forward [taps][rows][Cin] -> [taps][rows][3 Cin], dgrad [taps][Cin][rows] ->
    [taps][Cin][3 rows]), refreshed when the parameters change (the same epochs PackedWeight.get keys on)."""
    arena = getattr(weights[0], "_nrpn_arena", None) if len(weights) == 1 else None
    key = (wp.data_ptr(), wpd.data_ptr() if wpd is not None else 0, _weight_epoch, arena[0].epoch if arena is not None else pack.key)
    ent = pack.__dict__.get("x3")
    if ent is None or ent[0] != key:
        w3 = split3(wp, _W_I)[0]
        if wpd is not None:
            _wait_dgrad_operands()       # the trainer refreshes the dgrad operands on the weight-gradient stream (ArenaWeights.prefetch_dgrad): this
                                         # stream reads them here, in the FORWARD pass, before the first dgrad's own wait
        wd3 = split3(wpd, _W_I)[0] if wpd is not None else None
        ent = (key, w3, wd3)
        pack.__dict__["x3"] = ent
    return ent[1], ent[2]


def column_sum_f32(t, out=None, accumulate=False):
    """f32 [rows, C] -> f32 [C] column sums, deterministic; ``out`` += when ``accumulate``.  Shapes the BatchNorm statistics reduction takes
    (C % 4 == 0, C / 4 dividing 256: every conv width of the VGG / ResNet paths) go through it -- 4-channel lanes over ~1000 row slabs, fp64
    finish: mean * rows -- the rest (e.g. C = 96) through nrpn_column_sum_f32."""
    t = t.contiguous()
    rows, c = t.shape
    tile = min(c, 1024)
    if c % 4 == 0 and 256 % (tile // 4) == 0 and c % tile == 0:
        mean, var = torch.empty(c, dtype=torch.float32, device=t.device), torch.empty(c, dtype=torch.float32, device=t.device)
        ws = torch.empty(query("bn_workspace_bytes", rows, c), dtype=torch.uint8, device=t.device)
        call("bn_stats", _p(t), rows, c, F32, _p(mean), _p(var), 0, 0, 0.1, _p(ws), _s())
        res = mean * float(rows)
        if out is None:
            return res
        return out.add_(res) if accumulate else out.copy_(res)
    if out is None:
        out = torch.empty(c, dtype=torch.float32, device=t.device)
    ws = torch.empty(query("column_sum_workspace_bytes", rows, c), dtype=torch.uint8, device=t.device)
    call("column_sum_f32", _p(t), rows, c, _p(out), 1 if accumulate else 0, _p(ws), _s())
    return out


def _seg_dims(segs):
    import ctypes
    flat = [int(v) for d in segs for v in d]
    return (ctypes.c_int32 * len(flat))(*flat)


def _conv_fwd(x, wp, bias, cout, wrows, ksize, flags, out_dtype, segs=None, mask=None, stats=None, scale=None, tile=0, halo_pairing=0):
    """segs: None, or the (X, Y, Z) dims of the grids laid end to end in x = [1, sum(X*Y*Z), 1, 1, C] (ragged list).
    stats: None, or a dict that asks for BatchNorm statistics out of the conv epilogue: when the shape's kernel has them, the launch
    fills stats['partials'] = f32 [P, 2, cout] (see nrpn_conv3d_fwd_stats); otherwise the dict stays empty.
    scale: None or f32 [cout]: y = acc * scale + bias (eval-mode BatchNorm folded into the conv, nrpn_conv_opts.scale).
    tile: per-call kernel selection (lib.TILE_*; 0 = the library's choice for the shape); halo_pairing: nrpn_conv_opts.halo_pairing (0 = default,
    1 = taps paired across chunk boundaries, 2 = 14 K-steps per chunk)."""
    import ctypes
    n, gx, gy, gz, cin = x.shape
    y = torch.empty((n, gx, gy, gz, cout), dtype=out_dtype, device=x.device)
    if out_dtype == torch.float32 and x.dtype == torch.bfloat16:
        flags |= CONV_OUT_F32
    if bias is not None:
        flags |= CONV_BIAS
    if segs is not None and ksize != 1:
        if scale is not None or mask is not None or tile:
            raise lib.NrpnError("ragged conv launches take no per-call options")
        wsb = query("conv3d_fwd_workspace_bytes", n, gx, gy, gz, cin, cout, ksize, _dt(x))
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
        dims = _seg_dims(segs)
        call("conv3d_fwd_ragged", _p(x), _p(wp), _p(bias), _p(y), len(segs), ctypes.addressof(dims), cin, cout, wrows, ksize, _dt(x), flags,
             _p(ws), _s())
        return y
    opts = lib.ConvOpts(tile=tile or CONV_TILE[0], scale=_p(scale), relu_mask=_p(mask), halo_pairing=halo_pairing or CONV_HALO_PAIRING[0])
    if stats is not None and segs is None and mask is None and out_dtype == x.dtype and wrows == cout:
        rows = _query_opts("conv3d_fwd_stats_rows_ex", opts, n, gx, gy, gz, cin, cout, ksize, _dt(x))
        if rows > 0:
            part = torch.empty((rows, 2, cout), dtype=torch.float32, device=x.device)
            opts.stats = part.data_ptr()
            call("conv3d_fwd_ex", _p(x), _p(wp), _p(bias), _p(y), n, gx, gy, gz, cin, cout, wrows, ksize, _dt(x), flags, 0, opts.ptr(), _s())
            stats["partials"] = part
            return y
    wsb = _query_opts("conv3d_fwd_workspace_bytes_ex", opts, n, gx, gy, gz, cin, cout, ksize, _dt(x))
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    call("conv3d_fwd_ex", _p(x), _p(wp), _p(bias), _p(y), n, gx, gy, gz, cin, cout, wrows, ksize, _dt(x), flags, _p(ws), opts.ptr(), _s())
    return y


# Tile override of every forward / dgrad launch of this process ([0] = lib.TILE_*; 0 = the library's per-shape choice).  A per-call
# nrpn_conv_opts field underneath -- NOT a library global -- so it is safe next to other threads; bench.py / tools set it for A/B runs.
CONV_TILE = [int(_os.environ.get("NRPN_CONV_TILE", "0"))]
CONV_HALO_PAIRING = [int(_os.environ.get("NRPN_HALO_PAIRING", "0"))]      # A/B: 0 = library default, 1 = cross-chunk tap pairing, 2 = off


# wgrad workspace = [27-bit tap mask per voxel (k3) | per-slice bias partials]; the masks depend only on the grid, so one workspace per
# (grid shape, stream) is kept and the masks are built once (NRPN_WGRAD_MASK_READY afterwards); the bias partials are consumed by the
# same C call that writes them, or by the reduce_slices launch that follows it on the same stream.
_WGRAD_WS = {}


def _wgrad_workspace(device, key, nbytes):
    # keyed by the stream the wgrad is enqueued on as well: a layer with a frozen parameter keeps its wgrad on the main stream while
    # arena layers of the same grid shape run on the side stream -- their bias partials (and the one-time mask build) must not share
    # a buffer across streams
    key = (device.index, _s()) + key
    ent = _WGRAD_WS.get(key)
    if ent is None or ent[0].numel() < nbytes:
        ent = [torch.empty(nbytes, dtype=torch.uint8, device=device), False]
        _WGRAD_WS[key] = ent
    ready, ent[1] = ent[1], True
    return ent[0], ready


# Layer-chain hints for ConvFn (passed as relu=(bool, flags) by modules that own a conv+ReLU -> conv chain, e.g. RPNHead):
#   CHAIN_MASK_INPUT_GRAD  this conv is the ONLY consumer of its input, and the input is the output of a fused conv+ReLU
#   CHAIN_GRAD_PREMASKED   this conv+ReLU's output goes only to a conv with CHAIN_MASK_INPUT_GRAD: the incoming gradient is
#                          already multiplied by the ReLU mask, so the separate relu_backward pass is skipped
CHAIN_MASK_INPUT_GRAD, CHAIN_GRAD_PREMASKED = 1, 2


IN_BACKWARD = [0]       # > 0 while ConvFn.backward is enqueuing (bench.py tells forward launches from dgrad launches of the same entry point)


class ConvFn(torch.autograd.Function):
    """Conv3d k in {1,3}, stride 1, 'same' padding, channels-last, optional fused bias + ReLU.
    ``weights``: one or more reference-layout parameters that share the GEMM (their rows are concatenated, then padded to
    ``rows_total``); the output has rows_total channels."""

    @staticmethod
    def forward(ctx, x, pack, rows_total, relu, out_f32, nw, *wb):
        segs = None
        chain = 0
        stats = None
        affine = None
        if isinstance(relu, tuple):        # (relu, chain[, stats holder[, affine]]): see CHAIN_* below and _conv_fwd
            relu, chain, *rest = relu
            stats = rest[0] if rest else None
            affine = rest[1] if len(rest) > 1 else None
        if isinstance(nw, tuple):          # (nw, segs): ragged voxel list, x = [1, sum voxels, 1, 1, C]
            nw, segs = nw
        weights, biases = wb[:nw], wb[nw:]
        _chk(x)
        ksize = weights[0].shape[2] if weights[0][0].numel() != x.shape[-1] else 1
        need_dgrad = x.requires_grad
        wp, wpd = pack.get(weights, x.dtype, rows_total, need_dgrad, x.shape[-1])
        bias = None
        if biases[0] is not None:
            if nw == 1 and biases[0].numel() == rows_total:
                bias = biases[0].detach().float().contiguous()
            else:       # rows of a fused multi-weight GEMM, zero-padded: assembled once per parameter update, not once per pyramid level
                bkey = tuple((b.data_ptr(), b._version) for b in biases) + (rows_total, _weight_epoch)
                if getattr(pack, "bias_key", None) != bkey:
                    parts = [b.detach().float().reshape(-1) for b in biases]
                    used = sum(p.numel() for p in parts)
                    if used < rows_total:
                        parts.append(parts[0].new_zeros(rows_total - used))
                    pack.bias, pack.bias_key = torch.cat(parts), bkey
                bias = pack.bias
        out_dtype = torch.float32 if out_f32 else x.dtype
        x3 = _x3_ok(x, ksize, segs, rows_total, any(ctx.needs_input_grad))
        xin = x
        if x3:      # bf16x3: split operands on the bf16 MFMA kernels, fp32 rows out; BatchNorm statistics then come from their own pass
            wp, wpd = _x3_weights(pack, weights, wp, wpd)
            xin, stats = split3(x, _ACT_I)[0], None
        if affine is not None:
            # eval-mode BatchNorm folded into this conv: y = acc * scale + shift (the conv's own bias is inside `shift`); forward only --
            # the HIP path has no eval-mode BatchNorm backward either (BatchNormFn.backward)
            scale, shift = affine
            y = _conv_fwd(xin, wp, shift, rows_total, rows_total, ksize, CONV_RELU if relu else 0, out_dtype, segs, None, None, scale)
            ctx.mark_non_differentiable(y)       # hip_nn._can_fold only folds when nothing upstream needs a gradient
            return y
        y = _conv_fwd(xin, wp, bias, rows_total, rows_total, ksize, CONV_RELU if relu else 0, out_dtype, segs, None, stats)
        ctx.save_for_backward(x, y if relu else None, wpd, *weights)
        ctx.x3 = x3
        ctx.meta = (rows_total, relu, nw, ksize, biases[0] is not None, segs, chain)
        ctx.sinks = ([_sink(w) for w in weights], [_sink(b) for b in biases])
        return y

    @staticmethod
    def backward(ctx, dy):
        IN_BACKWARD[0] += 1
        try:
            return ConvFn._backward(ctx, dy)
        finally:
            IN_BACKWARD[0] -= 1

    @staticmethod
    def _backward(ctx, dy):
        x, y, wpd, *weights = ctx.saved_tensors
        rows_total, relu, nw, ksize, has_bias, segs, chain = ctx.meta
        n, gx, gy, gz, cin = x.shape
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if relu and not (chain & CHAIN_GRAD_PREMASKED):
            dyr = torch.empty_like(dy)
            yy = y if y.dtype == dy.dtype else y.to(dy.dtype)
            call("relu_backward", _p(yy), _p(dy), _p(dyr), dy.numel(), _dt(dy), _s())
            dy = dyr
        dx = None
        x3 = getattr(ctx, "x3", False)
        dyp = None
        if x3 and segs is not None:
            raise lib.NrpnError("bf16x3 backward of a ragged conv: the split dgrad / wgrad launches take no segments")
        if x3:
            # bf16x3: one pass over dy writes the interleaved operand of the dgrad launch and the planes of the wgrad launch
            dys, dyp = split3(dy, _ACT_I if ctx.needs_input_grad[0] else None, 3, _DY_P)
        if ctx.needs_input_grad[0]:
            _wait_dgrad_operands()
            # CHAIN_MASK_INPUT_GRAD: x is the ReLU output of the layer that receives dx and this conv is its only consumer, so that
            # layer's ReLU backward is applied in this dgrad's epilogue (dx = 0 where x <= 0) instead of a separate pass
            mask = x if (chain & CHAIN_MASK_INPUT_GRAD) and segs is None else None
            if x3:
                dx = _conv_fwd(dys, wpd, None, cin, cin, ksize, 0, torch.float32)
                if mask is not None:          # fp32 rows out of bf16 operands: the epilogue takes no mask, one elementwise pass instead
                    dxm = torch.empty_like(dx)
                    call("relu_backward", _p(x), _p(dx), _p(dxm), dx.numel(), _dt(dx), _s())
                    dx = dxm
            else:
                dx = _conv_fwd(dy, wpd, None, cin, cin, ksize, 0, x.dtype, segs, mask)
        taps = ksize ** 3
        wsinks, bsinks = ctx.sinks
        side = _wgrad_side_stream(x.device) if all(k is not None for k in wsinks) and (not has_bias or all(k is not None for k in bsinks)) else None
        wgrad = (lambda: ConvFn._wgrad_x3(ctx, x, dy, dyp, weights)) if x3 else (lambda: ConvFn._wgrad(ctx, x, dy, weights))
        if side is None:
            return (dx, None, None, None, None, None, *wgrad())
        main = torch.cuda.current_stream(x.device)
        side.wait_stream(main)              # dy (after the ReLU mask) is ready; also orders this wgrad behind the arena's zero fill
        x.record_stream(side)
        dy.record_stream(side)
        if dyp is not None:
            dyp.record_stream(side)
        _WGRAD_SIDE["keep"].append((x, dy, dyp))
        with torch.cuda.stream(side):       # every gradient goes straight into the arena here: nothing is handed back to autograd
            res = wgrad()
        if not _WGRAD_SIDE["dirty"]:        # first side-stream wgrad of this backward pass: join when the pass ends
            _WGRAD_SIDE["dirty"] = True
            torch.autograd.Variable._execution_engine.queue_callback(wgrad_stream_join)
        return (dx, None, None, None, None, None, *res)

    @staticmethod
    def _wgrad_x3(ctx, x, dy, dyp, weights):
        """bf16x3 weight gradient: dW = dy_hi (x) x_hi + dy_hi (x) x_lo + dy_lo (x) x_hi as ONE bf16 wgrad launch over three "scenes"
        (planes stacked on the batch axis, which the kernel sums over); the bias gradient is the fp32 column sum of dy."""
        rows_total, relu, nw, ksize, has_bias, segs, chain = ctx.meta
        n, gx, gy, gz, cin = x.shape
        taps = ksize ** 3
        wsinks, bsinks = ctx.sinks
        xp = split3(x, None, 3, _X_P)[1]
        n3 = 3 * n
        slices = query("conv3d_wgrad_slices", n3, gx, gy, gz, cin, rows_total, rows_total, ksize, BF16)
        gwp = torch.empty((slices, taps, rows_total, cin), dtype=torch.float32, device=x.device)
        ws, mask_ready = _wgrad_workspace(x.device, (n3, gx, gy, gz, ksize, None),
                                          query("conv3d_wgrad_workspace_bytes", n3, gx, gy, gz, cin, rows_total, rows_total, ksize, BF16))
        call("conv3d_wgrad", _p(xp), _p(dyp), _p(gwp), 0, n3, gx, gy, gz, cin, rows_total, rows_total, ksize, BF16, 2 if mask_ready else 0, _p(ws), _s())
        res = _deliver_wgrad(weights, wsinks, bsinks, gwp, None, 0, slices, rows_total, cin, taps, False, False, False)
        gbs = _deliver_bias_x3(dy.reshape(-1, rows_total), weights, bsinks) if has_bias else tuple([None] * nw)
        return (*res[:nw], *gbs)

    @staticmethod
    def _wgrad(ctx, x, dy, weights):
        rows_total, relu, nw, ksize, has_bias, segs, chain = ctx.meta
        n, gx, gy, gz, cin = x.shape
        taps = ksize ** 3
        wsinks, bsinks = ctx.sinks
        slices = query("conv3d_wgrad_slices", n, gx, gy, gz, cin, rows_total, rows_total, ksize, _dt(x))
        gwp = torch.empty((slices, taps, rows_total, cin), dtype=torch.float32, device=x.device)     # per-slice partials, summed by the unpack
        direct_bias = has_bias and nw == 1 and bsinks[0] is not None
        gb = bsinks[0].slot if direct_bias else (torch.empty(rows_total, dtype=torch.float32, device=x.device) if has_bias else None)
        ws, mask_ready = _wgrad_workspace(x.device, (n, gx, gy, gz, ksize, segs),
                                          query("conv3d_wgrad_workspace_bytes", n, gx, gy, gz, cin, rows_total, rows_total, ksize, _dt(x)))
        # arena path: weight-slice sum and bias-slice sum of this layer go out as ONE launch (reduce_slices), so the wgrad call leaves
        # the bias partials in the workspace
        fused_reduce = nw == 1 and wsinks[0] is not None and wsinks[0].flat is not None and rows_total == weights[0].shape[0]
        defer_bias = fused_reduce and direct_bias
        wflags = int(direct_bias) | (2 if mask_ready else 0) | (4 if defer_bias else 0)
        if segs is None or ksize == 1:
            call("conv3d_wgrad", _p(x), _p(dy), _p(gwp), _p(gb), n, gx, gy, gz, cin, rows_total, rows_total, ksize, _dt(x), wflags, _p(ws), _s())
        else:
            import ctypes
            dims = _seg_dims(segs)
            call("conv3d_wgrad_ragged", _p(x), _p(dy), _p(gwp), _p(gb), len(segs), ctypes.addressof(dims), cin, rows_total, rows_total, ksize,
                 _dt(x), wflags, _p(ws), _s())
        bias_part = ws.data_ptr() + query("conv3d_wgrad_bias_offset", n, gx, gy, gz, ksize) if defer_bias else 0
        return _deliver_wgrad(weights, wsinks, bsinks, gwp, gb, bias_part, slices, rows_total, cin, taps, has_bias, direct_bias, defer_bias)


def _deliver_wgrad(weights, wsinks, bsinks, gwp, gb, bias_part, slices, rows_total, cin, taps, has_bias, direct_bias, defer_bias):
    """Sum the per-slice partial gradients ``gwp`` [slices, taps, rows_total, cin] of a (possibly fused multi-weight) GEMM into their
    destinations: the flat gradient arena (GradSink) or fresh tensors handed back to autograd.  -> (*weight grads, *bias grads), None where
    a sink took the gradient."""
    nw = len(weights)
    gws, gbs, row = [], [], 0
    for i, w in enumerate(weights):
        if wsinks[i] is not None and wsinks[i].flat is not None and nw == 1 and rows_total == w.shape[0]:
            # the arena keeps this gradient in the partials' own layout: ordered sum of the slices, added in place
            call("reduce_slices", _p(gwp), slices, gwp[0].numel(), _p(wsinks[i].flat), 1, bias_part if defer_bias else 0, rows_total, rows_total,
                 _p(gb) if defer_bias else 0, 1, _s())
            wsinks[i].notify()
            gws.append(None)
        elif wsinks[i] is not None:
            if not wsinks[i].slot.is_contiguous():
                raise lib.NrpnError("a GEMM-layout arena weight reached a fused multi-weight GEMM (mark the module _nrpn_fused_gemm)")
            call("unpack_conv_wgrad", _p(gwp), w.shape[0], cin, taps, rows_total, row, _p(wsinks[i].slot), 1, slices, _s())
            wsinks[i].notify()
            gws.append(None)
        else:
            gw = torch.empty_like(w, dtype=torch.float32)
            call("unpack_conv_wgrad", _p(gwp), w.shape[0], cin, taps, rows_total, row, _p(gw), 0, slices, _s())
            gws.append(gw)
        if not has_bias:
            gbs.append(None)
        elif direct_bias:
            bsinks[0].notify()
            gbs.append(None)
        elif bsinks[i] is not None:
            bsinks[i].slot.add_(gb[row:row + w.shape[0]])
            bsinks[i].notify()
            gbs.append(None)
        else:
            gbs.append(gb[row:row + w.shape[0]].clone())
        row += w.shape[0]
    return (*gws, *gbs)


def _deliver_bias_x3(dy2d, weights, bsinks):
    """Bias gradients of a (possibly fused multi-weight) GEMM from the fp32 column sums of dy: into the sinks, or handed back."""
    gb = column_sum_f32(dy2d)
    out, row = [], 0
    for i, w in enumerate(weights):
        part = gb[row:row + w.shape[0]]
        if bsinks[i] is not None:
            bsinks[i].slot.add_(part)
            bsinks[i].notify()
            out.append(None)
        else:
            out.append(part.clone())
        row += w.shape[0]
    return tuple(out)


def _on_wgrad_stream(device, tensors, sinks_complete, fn):
    """Run ``fn()`` (a weight-gradient launch sequence whose results all go into GradSinks) on the weight-gradient side stream behind the
    current stream, or inline when the side stream is off / a gradient has to be handed back to autograd."""
    side = _wgrad_side_stream(device) if sinks_complete else None
    if side is None:
        return fn()
    main = torch.cuda.current_stream(device)
    side.wait_stream(main)
    for t in tensors:
        t.record_stream(side)
    _WGRAD_SIDE["keep"].append(tuple(tensors))
    with torch.cuda.stream(side):
        res = fn()
    if not _WGRAD_SIDE["dirty"]:
        _WGRAD_SIDE["dirty"] = True
        torch.autograd.Variable._execution_engine.queue_callback(wgrad_stream_join)
    return res


# ======================================================================================================================
# sampled-anchor cones of the RPN head (training)
# ======================================================================================================================
CONE_ENABLED = [_os.environ.get("NRPN_CONE", "1") != "0"]      # A/B switch: cone evaluation of the RPN head in training; default on


class ConePlan:
    """Sorted voxel lists S_0 c ... c S_depth of one training step (csrc/cone.hip) over the ragged (level-major, scene-major) voxel space
    of the pyramid: ``lists`` int32 [depth + 1, total, 2] on the device, ``counts`` on the host once the sampler's read-back has happened."""

    def __init__(self, grids, n_scenes, depth, device):
        import ctypes
        self.grids = [tuple(int(v) for v in g) for g in grids]
        self.n, self.depth = int(n_scenes), int(depth)
        self.segs = [g for g in self.grids for _ in range(self.n)]
        self.nseg = len(self.segs)
        self.dims = _seg_dims(self.segs)
        self.dims_ptr = ctypes.addressof(self.dims)
        self.cells = [g[0] * g[1] * g[2] for g in self.grids]
        self.level_start = [0]
        for c in self.cells:
            self.level_start.append(self.level_start[-1] + c * self.n)
        self.total = self.level_start[-1]
        self.lists = torch.empty((self.depth + 1, self.total, 2), dtype=torch.int32, device=device)
        self.counts_dev = torch.empty(self.depth + 2, dtype=torch.int32, device=device)
        self.counts = None

    def finish(self, host):
        if host[self.depth + 1]:
            raise RuntimeError("cone_build: a sampled anchor index lies outside the anchor pyramid")
        self.counts = [int(v) for v in host[:self.depth + 1]]

    def rows(self, k):
        return self.lists[k].data_ptr(), self.counts[k]

    def tensors(self):
        return [self.lists, self.counts_dev]


def cone_from_indices(plan, pos, neg, table):
    """Build ``plan`` from flat batch indices (scene * T + anchor), e.g. a test's injected sample; synchronises (sizes of boolean selections)."""
    T, n = table.total, plan.n
    ps = [(pos[(pos >= i * T) & (pos < (i + 1) * T)] - i * T).long() for i in range(n)]
    ns = [(neg[(neg >= i * T) & (neg < (i + 1) * T)] - i * T).long() for i in range(n)]
    mp, mn = max(1, max(p.numel() for p in ps)), max(1, max(q.numel() for q in ns))
    dev = plan.lists.device
    out_pos = torch.zeros((n, mp), dtype=torch.int64, device=dev)
    out_neg = torch.zeros((n, mn), dtype=torch.int64, device=dev)
    cnt = []
    for i in range(n):
        out_pos[i, :ps[i].numel()] = ps[i]
        out_neg[i, :ns[i].numel()] = ns[i]
        cnt += [ps[i].numel(), ns[i].numel(), 0]
    counts = torch.tensor(cnt, dtype=torch.int32, device=dev)
    plan.finish(cone_build(plan, out_pos, out_neg, counts, table).cpu().tolist())
    return plan


def cone_build(plan, out_pos, out_neg, counts, table):
    """Enqueue the list build of ``plan`` on the current stream from the sampler's raw output (see sample_pos_neg); no synchronisation."""
    import ctypes
    if table.A * sum(plan.cells) != table.total or len(table.counts) != len(plan.grids):
        raise lib.NrpnError("cone_build: the anchor table and the feature grids disagree")
    offs = (ctypes.c_int64 * (len(table.offsets)))(*[int(v) for v in table.offsets])
    ws = torch.empty(query("cone_workspace_bytes", plan.total), dtype=torch.uint8, device=out_pos.device)
    call("cone_build", _p(out_pos), _p(out_neg), _p(counts), plan.n, out_pos.shape[1], out_neg.shape[1], len(plan.grids), ctypes.addressof(offs),
         table.A, plan.dims_ptr, plan.depth, _p(plan.lists), plan.total, _p(plan.counts_dev), _p(ws), _s())
    return plan.counts_dev


def conv_rows_fwd(x, wp, bias, y, plan, k, cin, cout, wrows, ksize, flags, mask=None):
    """Forward / dgrad of a k1 / k3 conv on the rows of list S_k only: x [total, cin], y [total, cout] (rows outside the list untouched)."""
    rows, nrows = plan.rows(k)
    if nrows == 0:
        return y
    if y.dtype == torch.float32 and x.dtype == torch.bfloat16:
        flags |= CONV_OUT_F32
    if bias is not None:
        flags |= CONV_BIAS
    wsb = lib.query("conv3d_fwd_rows_workspace_bytes", nrows, cin, cout, ksize, _dt(x))      # (not memoised: the list length changes every step)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    call("conv3d_fwd_rows", _p(x), _p(wp), _p(bias), _p(y), rows, nrows, plan.nseg, plan.dims_ptr, cin, cout, wrows, ksize, _dt(x), flags, _p(mask),
         _p(ws), _s())
    return y


def conv_rows_wgrad(x, dy, weights, biases, rows_total, ksize, plan, k, sinks=None):
    """Weight (+ bias) gradient of a conv from the rows of S_k only (dy is zero elsewhere); delivery as ConvFn._wgrad.
    ``sinks``: (weight sinks, bias sinks) looked up by the caller on the original Parameter objects; default: on ``weights`` / ``biases``."""
    rows, nrows = plan.rows(k)
    cin = x.shape[-1]
    taps = ksize ** 3
    nw = len(weights)
    wsinks, bsinks = sinks if sinks is not None else ([_sink(w) for w in weights], [_sink(b) for b in biases])
    has_bias = biases[0] is not None
    if nrows == 0:      # nothing sampled: zero gradients (sinks are only told that this use of the parameter is done)
        out = []
        for t, sk in list(zip(weights, wsinks)) + list(zip(biases, bsinks)):
            if t is None:
                out.append(None)
            elif sk is not None:
                sk.notify()
                out.append(None)
            else:
                out.append(torch.zeros_like(t, dtype=torch.float32))
        return tuple(out)
    slices = lib.query("conv3d_wgrad_slices", 1, nrows, 1, 1, cin, rows_total, rows_total, ksize, _dt(x))      # (not memoised: nrows changes every step)
    gwp = torch.empty((slices, taps, rows_total, cin), dtype=torch.float32, device=x.device)
    direct_bias = has_bias and nw == 1 and bsinks[0] is not None
    gb = bsinks[0].slot if direct_bias else (torch.empty(rows_total, dtype=torch.float32, device=x.device) if has_bias else None)
    ws = torch.empty(max(256, slices * rows_total * 4), dtype=torch.uint8, device=x.device)
    fused_reduce = nw == 1 and wsinks[0] is not None and wsinks[0].flat is not None and rows_total == weights[0].shape[0]
    defer_bias = fused_reduce and direct_bias
    wflags = int(direct_bias) | (4 if defer_bias else 0)
    call("conv3d_wgrad_rows", _p(x), _p(dy), _p(gwp), _p(gb), rows, nrows, plan.nseg, plan.dims_ptr, cin, rows_total, rows_total, ksize, _dt(x),
         wflags, _p(ws), _s())
    return _deliver_wgrad(weights, wsinks, bsinks, gwp, gb, ws.data_ptr(), slices, rows_total, cin, taps, has_bias, direct_bias, defer_bias)


def conv_rows_wgrad_x3(x, dy, weights, biases, rows_total, ksize, plan, k, sinks=None):
    """bf16x3 form of conv_rows_wgrad (one weight): the three split products dy_hi (x) x_hi, dy_hi (x) x_lo, dy_lo (x) x_hi as three bf16
    row-list launches whose slice partials are summed together by the usual delivery; bias gradient = fp32 column sum of dy (zero outside
    the list)."""
    rows, nrows = plan.rows(k)
    if nrows == 0 or len(weights) != 1:
        return conv_rows_wgrad(x, dy, weights, biases, rows_total, ksize, plan, k, sinks)
    cin = x.shape[-1]
    taps = ksize ** 3
    wsinks, bsinks = sinks if sinks is not None else ([_sink(w) for w in weights], [_sink(b) for b in biases])
    has_bias = biases[0] is not None
    xpl = split3(x, None, 2, 0b10)[1]          # [hi ; lo]
    dpl = split3(dy, None, 2, 0b10)[1]
    slices = lib.query("conv3d_wgrad_slices", 1, nrows, 1, 1, cin, rows_total, rows_total, ksize, BF16)
    gwp = torch.empty((3 * slices, taps, rows_total, cin), dtype=torch.float32, device=x.device)
    ws = torch.empty(max(256, slices * rows_total * 4), dtype=torch.uint8, device=x.device)
    for j, (di, xi) in enumerate(((0, 0), (0, 1), (1, 0))):
        call("conv3d_wgrad_rows", _p(xpl[xi]), _p(dpl[di]), gwp[j * slices].data_ptr(), 0, rows, nrows, plan.nseg, plan.dims_ptr, cin, rows_total,
             rows_total, ksize, BF16, 0, _p(ws), _s())
    res = _deliver_wgrad(weights, wsinks, bsinks, gwp, None, 0, 3 * slices, rows_total, cin, taps, False, False, False)
    gbs = _deliver_bias_x3(dy.reshape(-1, rows_total), weights, bsinks) if has_bias else (None,)
    return (res[0], *gbs)


class ConeHeadFn(torch.autograd.Function):
    """The whole RPN head in training -- conv_depth x [Conv3d k3 + ReLU] + the fused cls / bbox GEMM over all pyramid levels + the
    reference's flatten (anchor.py:RPNHead, rpn.py:105-130) -- evaluated on the sampled-anchor cones of ``plan`` only: layer i (0-based,
    D layers) is computed on S_{D-1-i}, the output GEMM on S_0; backward visits the same sets (weight gradients from the listed rows, input
    gradients on the next larger set, the first layer's input gradient densely per level for the FPN).  One autograd node, ~25 launches,
    instead of ~150 for the dense head at four levels.  logits / deltas are exact on the sampled anchors' voxels and ZERO elsewhere: only
    the training loss, which reads the sampled rows, may consume them (RegionProposalNetwork.forward decides)."""

    @staticmethod
    def forward(ctx, plan, head, A, dw, *tensors):
        L = len(plan.grids)
        feats = tensors[:L]
        convs = [m for m in head.conv if isinstance(m, torch.nn.Conv3d)]
        D = len(convs)
        cw, cb = tensors[L:L + D], tensors[L + D:L + 2 * D]
        ow, ob = tensors[L + 2 * D:L + 2 * D + 2], tensors[L + 2 * D + 2:L + 2 * D + 4]
        if plan.depth != max(D - 1, 0):
            raise lib.NrpnError("ConeHeadFn: the plan must hold the lists S_0 .. S_{conv_depth - 1}")
        n, C = feats[0].shape[0], feats[0].shape[-1]
        dt = feats[0].dtype
        dev = feats[0].device
        V = plan.total
        for f in feats:
            _chk(f)
        x0 = torch.cat([f.reshape(-1, C) for f in feats], dim=0)
        if x0.shape[0] != V:
            raise lib.NrpnError("ConeHeadFn: feature maps and cone plan disagree on the voxel count")
        from .model.hip_nn import _pack_of
        hs, ops_w = [x0], []
        need_dgrad = any(f.requires_grad for f in feats)
        x3 = SPLIT3[0] and dt == torch.float32 and C % 32 == 0      # bf16x3: the 3x3x3 layers multiply split operands (the output GEMM stays fp32)
        for i, cv in enumerate(convs):
            wp, wpd = _pack_of(cv).get([cw[i]], dt, C, True, C)
            bias = cb[i].detach().float().contiguous() if cb[i] is not None else None
            y = torch.empty((V, C), dtype=dt, device=dev)
            if x3:
                wp, wpd = _x3_weights(_pack_of(cv), [cw[i]], wp, wpd)
                # (rows outside the cone hold whatever the allocator left: split like the rest, never read by a listed row's taps)
                conv_rows_fwd(split3(hs[-1], _ACT_I)[0], wp, bias, y, plan, D - 1 - i, 3 * C, C, C, 3, CONV_RELU)
            else:
                conv_rows_fwd(hs[-1], wp, bias, y, plan, D - 1 - i, C, C, C, 3, CONV_RELU)
            hs.append(y)
            ops_w.append(wpd)
        rows_total = head.head_rows
        wpo, wpdo = head._pack.get(list(ow), dt, rows_total, True, C)
        bias_o = None
        if ob[0] is not None:
            pack = head._pack
            bkey = tuple((b.data_ptr(), b._version) for b in ob) + (rows_total, _weight_epoch)
            if getattr(pack, "bias_key", None) != bkey:
                parts = [b.detach().float().reshape(-1) for b in ob]
                used = sum(p.numel() for p in parts)
                if used < rows_total:
                    parts.append(parts[0].new_zeros(rows_total - used))
                pack.bias, pack.bias_key = torch.cat(parts), bkey
            bias_o = pack.bias
        out = torch.zeros((V, rows_total), dtype=torch.float32, device=dev)
        conv_rows_fwd(hs[-1], wpo, bias_o, out, plan, 0, C, rows_total, rows_total, 1, 0)
        T = sum(plan.cells) * A
        logits = torch.empty((n, T), dtype=torch.float32, device=dev)
        deltas = torch.empty((n, T, dw), dtype=torch.float32, device=dev)
        off = 0
        for l, c in enumerate(plan.cells):
            for i in range(n):
                r0 = plan.level_start[l] + i * c
                call("head_flatten_f32", _p(out[r0:r0 + c]), c, rows_total, A, dw, _p(logits[i, off:]), _p(deltas[i, off:]), _s())
            off += c * A
        ctx.save_for_backward(*hs, *ops_w, wpdo, *cw, *[b for b in cb if b is not None], *ow, *[b for b in ob if b is not None])
        ctx.meta = (plan, head, A, dw, L, D, n, C, dt, rows_total, [b is not None for b in cb], [b is not None for b in ob], need_dgrad)
        ctx.x3 = x3
        # gradient sinks are attributes of the Parameter objects: looked up here, on the objects the caller passed (as ConvFn does)
        ctx.sinks = ([_sink(w) for w in cw], [_sink(b) for b in cb], [_sink(w) for w in ow], [_sink(b) for b in ob])
        return logits, deltas

    @staticmethod
    def backward(ctx, g_logits, g_deltas):
        IN_BACKWARD[0] += 1
        try:
            return ConeHeadFn._backward(ctx, g_logits, g_deltas)
        finally:
            IN_BACKWARD[0] -= 1

    @staticmethod
    def _backward(ctx, g_logits, g_deltas):
        plan, head, A, dw, L, D, n, C, dt, rows_total, has_cb, has_ob, need_dgrad = ctx.meta
        saved = list(ctx.saved_tensors)
        hs, saved = saved[:D + 1], saved[D + 1:]
        wpds, saved = saved[:D], saved[D:]
        wpdo, saved = saved[0], saved[1:]
        cw, saved = saved[:D], saved[D:]
        cb = []
        for h in has_cb:
            cb.append(saved.pop(0) if h else None)
        ow, saved = saved[:2], saved[2:]
        ob = [saved.pop(0) if h else None for h in has_ob]
        dev = hs[0].device
        V = plan.total
        g_logits, g_deltas = g_logits.contiguous(), g_deltas.contiguous()
        dout = torch.empty((V, rows_total), dtype=dt, device=dev)
        off = 0
        for l, c in enumerate(plan.cells):
            for i in range(n):
                r0 = plan.level_start[l] + i * c
                call("head_unflatten", _p(g_logits[i, off:]), _p(g_deltas[i, off:]), c, rows_total, A, dw, 0, _p(dout[r0:r0 + c]), _dt(dout), _s())
            off += c * A
        _wait_dgrad_operands()

        cws, cbs, ows, obs = ctx.sinks

        def sinks_ok(ws_, bs_, bt_):
            return all(k_ is not None for k_ in ws_) and all(k_ is not None for k_, b in zip(bs_, bt_) if b is not None)

        # output GEMM: weight / bias gradients from the rows of S_0; input gradient on S_0 with the last layer's ReLU mask
        g_ow = _on_wgrad_stream(dev, [hs[D], dout, plan.lists], sinks_ok(ows, obs, ob),
                                lambda: conv_rows_wgrad(hs[D], dout, list(ow), list(ob), rows_total, 1, plan, 0, (ows, obs)))
        dh = torch.zeros((V, C), dtype=dt, device=dev)
        conv_rows_fwd(dout, wpdo, None, dh, plan, 0, rows_total, C, C, 1, 0, mask=hs[D] if D > 0 else None)
        g_cw, g_cb = [None] * D, [None] * D
        g_feats = [None] * L
        if D == 0 and need_dgrad:       # no hidden layers: the output GEMM reads the FPN maps directly
            for l in range(L):
                g = plan.grids[l]
                g_feats[l] = dh[plan.level_start[l]:plan.level_start[l + 1]].view(n, g[0], g[1], g[2], C)
        x3 = getattr(ctx, "x3", False)
        rows_wgrad = conv_rows_wgrad_x3 if x3 else conv_rows_wgrad
        for i in range(D - 1, -1, -1):
            k = D - 1 - i
            x_i, dy_i = hs[i], dh
            res = _on_wgrad_stream(dev, [x_i, dy_i, plan.lists], sinks_ok([cws[i]], [cbs[i]], [cb[i]]),
                                   lambda x_i=x_i, dy_i=dy_i, i=i, k=k: rows_wgrad(x_i, dy_i, [cw[i]], [cb[i]], C, 3, plan, k, ([cws[i]], [cbs[i]])))
            g_cw[i] = res[0]
            g_cb[i] = res[1] if len(res) > 1 else None
            dhs = split3(dh, _ACT_I)[0] if (x3 and (i > 0 or need_dgrad)) else None      # bf16x3: the dgrad launches read the split gradient
            if i > 0:
                dprev = torch.zeros((V, C), dtype=dt, device=dev)
                if x3:       # fp32 rows out of bf16 operands take no epilogue mask: the layer's ReLU backward is one elementwise pass
                    raw = torch.zeros((V, C), dtype=dt, device=dev)
                    conv_rows_fwd(dhs, wpds[i], None, raw, plan, k + 1, 3 * C, C, C, 3, 0)
                    call("relu_backward", _p(hs[i]), _p(raw), _p(dprev), raw.numel(), _dt(raw), _s())
                else:
                    conv_rows_fwd(dh, wpds[i], None, dprev, plan, k + 1, C, C, C, 3, 0, mask=hs[i])
                dh = dprev
            elif need_dgrad:
                # first layer: its input gradient goes to the FPN output convs of every level -- dense, per level (halo kernel on the
                # finest grid); dh is zero outside S_{D-1}
                for l, c in enumerate(plan.cells):
                    g = plan.grids[l]
                    if x3:
                        dyl = dhs[plan.level_start[l]:plan.level_start[l + 1]].view(n, g[0], g[1], g[2], 3 * C)
                        g_feats[l] = _conv_fwd(dyl, wpds[0], None, C, C, 3, 0, torch.float32)
                        continue
                    dyl = dh[plan.level_start[l]:plan.level_start[l + 1]].view(n, g[0], g[1], g[2], C)
                    g_feats[l] = _conv_fwd(dyl, wpds[0], None, C, C, 3, 0, dt)
        return (None, None, None, None, *g_feats, *g_cw, *g_cb, *g_ow)


STEM_HALO = [_os.environ.get("NRPN_STEM_HALO", "1") != "0"]      # A/B switch: halo-form stem forward (bf16, stride 2, even Z, Cout 64); default on


class StemFn(torch.autograd.Function):
    """Conv3d(4 -> C, k7, pad 3, stride s) on [N,X,Y,Z,4] (reference feature_extractor.py:336,341)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, cache, affine=None, relu=False):
        _chk(x)
        n, gx, gy, gz, _ = x.shape
        cout = weight.shape[0]
        halo = STEM_HALO[0] and bool(query("stem_halo_supported", gz, cout, stride, _dt(x)))
        key = (weight.data_ptr(), weight._version, x.dtype, _weight_epoch, halo)
        if cache.get("key") != key:
            PACK_COUNT["stem"] += 1
            wc = weight.detach().contiguous()
            if halo:      # bf16 [Cout][50 * 32]: (dx, dy) slot major, 8 z positions x 4 channels inside a slot
                wp = torch.empty((cout, query("stem_halo_kpad")), dtype=x.dtype, device=x.device)
                call("pack_stem_weight_halo", _p(wc), cout, _p(wp), _s())
            else:
                kpad = query("stem_kpad", _dt(x))
                wp = torch.empty((cout, kpad), dtype=x.dtype, device=x.device)
                call("pack_stem_weight", _p(wc), cout, _dt(x), _p(wp), _s())
            cache["key"], cache["wp"] = key, wp
        wp = cache["wp"]
        o = [(g - 1) // stride + 1 for g in (gx, gy, gz)]
        y = torch.empty((n, o[0], o[1], o[2], cout), dtype=x.dtype, device=x.device)
        if affine is not None:       # eval-mode BatchNorm (+ ReLU) folded into the stem: no-grad forward only
            scale, shift = affine
            if halo:
                call("conv3d_stem_fwd_halo", _p(x), _p(wp), _p(shift), _p(scale), _p(y), n, gx, gy, gz, cout, CONV_BIAS | (CONV_RELU if relu else 0), _s())
            else:
                opts = lib.ConvOpts(scale=_p(scale))
                call("conv3d_stem_fwd_ex", _p(x), _p(wp), _p(shift), _p(y), n, gx, gy, gz, cout, stride, _dt(x),
                     CONV_BIAS | (CONV_RELU if relu else 0), opts.ptr(), _s())
            ctx.mark_non_differentiable(y)
            return y
        b = bias.detach().float().contiguous() if bias is not None else None
        if halo:
            call("conv3d_stem_fwd_halo", _p(x), _p(wp), _p(b), 0, _p(y), n, gx, gy, gz, cout, CONV_BIAS if b is not None else 0, _s())
        else:
            call("conv3d_stem_fwd", _p(x), _p(wp), _p(b), _p(y), n, gx, gy, gz, cout, stride, _dt(x), CONV_BIAS if b is not None else 0, _s())
        ctx.save_for_backward(x, weight)
        ctx.meta = (stride, bias is not None)
        ctx.sinks = (_sink(weight), _sink(bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, has_bias = ctx.meta
        dy = dy.contiguous()
        n, gx, gy, gz, _ = x.shape
        cout = weight.shape[0]
        kpad = query("stem_kpad", _dt(x))
        slices = query("stem_wgrad_slices", n, gx, gy, gz, cout, stride, _dt(x))
        per = query("stem_wgrad_slice_floats", n, gx, gy, gz, cout, stride, _dt(x))
        gwp = torch.empty((slices, per), dtype=torch.float32, device=x.device)
        wsink, bsink = ctx.sinks
        direct_bias = has_bias and bsink is not None
        gb = bsink.slot if direct_bias else (torch.empty(cout, dtype=torch.float32, device=x.device) if has_bias else None)
        ws = torch.empty(query("stem_wgrad_workspace_bytes", n, gx, gy, gz, cout, stride, _dt(x)), dtype=torch.uint8, device=x.device) if has_bias else None
        call("conv3d_stem_wgrad", _p(x), _p(dy), _p(gwp), _p(gb), n, gx, gy, gz, cout, stride, _dt(x), int(direct_bias), _p(ws), _s())
        if wsink is not None:
            call("unpack_stem_wgrad", _p(gwp), cout, _dt(x), _p(wsink.slot), 1, slices, per, _s())
            wsink.notify()
            gw = None
        else:
            gw = torch.empty_like(weight, dtype=torch.float32)
            call("unpack_stem_wgrad", _p(gwp), cout, _dt(x), _p(gw), 0, slices, per, _s())
        if direct_bias:
            bsink.notify()
            gb = None
        return None, gw, gb, None, None, None, None


BN_STATS_EPOCH = [0]      # bumped by every training-mode BatchNorm forward (running statistics rewritten behind torch's back)


class BatchNormFn(torch.autograd.Function):
    """BatchNorm3d (+ fused ReLU) on channels-last rows; training uses per-rank batch statistics (no SyncBN, as the reference)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, training, momentum, eps, relu, partials=None):
        _chk(x)
        c = x.shape[-1]
        rows = x.numel() // c
        dev = x.device
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        if training:
            # the statistics kernels update running_mean / running_var through raw pointers (no tensor._version bump): folded eval-mode
            # (scale, shift) pairs cached on the buffer versions would go stale -- hip_nn.bn_fold keys on this epoch as well
            BN_STATS_EPOCH[0] += 1
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            var = torch.empty(c, dtype=torch.float32, device=dev)
            if partials is not None:      # the producing conv left (sum, sum of squares) partials of x in its epilogue: only finish them
                call("bn_stats_finalize", _p(partials), partials.shape[0], rows, c, _p(mean), _p(var), _p(rmean), _p(rvar), float(momentum), _s())
            else:
                ws = torch.empty(query("bn_workspace_bytes", rows, c), dtype=torch.uint8, device=dev)
                call("bn_stats", _p(x), rows, c, _dt(x), _p(mean), _p(var), _p(rmean), _p(rvar), float(momentum), _p(ws), _s())
        else:
            mean, var = rmean.float().contiguous(), rvar.float().contiguous()
        y = torch.empty_like(x)
        call("bn_apply", _p(x), _p(y), rows, c, _dt(x), _p(mean), _p(var), _p(g32), _p(b32), float(eps), int(relu), _s())
        ctx.save_for_backward(x, mean, var, g32, b32)      # y is not kept: the backward recomputes the ReLU mask from x
        ctx.meta = (training, eps, relu)
        ctx.sinks = (_sink(gamma), _sink(beta))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, var, g32, b32 = ctx.saved_tensors
        training, eps, relu = ctx.meta
        if not training:
            raise lib.NrpnError("BatchNorm backward in eval mode is not supported by the HIP path")
        dy = dy.contiguous()
        c = x.shape[-1]
        rows = x.numel() // c
        dev = x.device
        dx = torch.empty_like(x)
        dgamma = torch.empty(c, dtype=torch.float32, device=dev)
        dbeta = torch.empty(c, dtype=torch.float32, device=dev)
        ws = torch.empty(query("bn_workspace_bytes", rows, c), dtype=torch.uint8, device=dev)
        gsink, bsink = ctx.sinks
        call("bn_backward", _p(x), 0, _p(dy), _p(dx), rows, c, _dt(x), _p(mean), _p(var), _p(g32), _p(b32), float(eps), int(relu), _p(dgamma),
             _p(dbeta), _p(gsink.slot) if gsink is not None else 0, _p(bsink.slot) if bsink is not None else 0, _p(ws), _s())
        if gsink is not None:
            gsink.notify()
            dgamma = None
        if bsink is not None:
            bsink.notify()
            dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p, ceil_mode):
        _chk(x)
        n, gx, gy, gz, c = x.shape
        o = [query("pool_out_size", g, k, s, p, int(ceil_mode)) for g in (gx, gy, gz)]
        y = torch.empty((n, o[0], o[1], o[2], c), dtype=x.dtype, device=x.device)
        arg = torch.empty(y.shape, dtype=torch.int8, device=x.device) if x.requires_grad else None
        call("maxpool3d_fwd", _p(x), _p(y), _p(arg), n, gx, gy, gz, c, k, s, p, int(ceil_mode), _dt(x), _s())
        ctx.save_for_backward(arg)
        ctx.meta = (x.shape, k, s, p, ceil_mode)
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        shape, k, s, p, ceil_mode = ctx.meta
        dy = dy.contiguous()
        n, gx, gy, gz, c = shape
        dx = torch.empty(shape, dtype=dy.dtype, device=dy.device)
        call("maxpool3d_bwd", _p(dy), _p(arg), _p(dx), n, gx, gy, gz, c, k, s, p, int(ceil_mode), _dt(dy), _s())
        return dx, None, None, None, None


class UpsampleAddFn(torch.autograd.Function):
    """fine += nearest_upsample(coarse), in place on ``fine`` (FPN top-down, reference fpn.py:150-155)."""

    @staticmethod
    def forward(ctx, fine, coarse):
        _chk(fine, coarse)
        n, fx, fy, fz, c = fine.shape
        _, cx, cy, cz, _ = coarse.shape
        call("upsample_add_fwd", _p(fine), _p(coarse), n, fx, fy, fz, cx, cy, cz, c, _dt(fine), _s())
        ctx.mark_dirty(fine)
        ctx.meta = (fine.shape, coarse.shape)
        return fine

    @staticmethod
    def backward(ctx, d):
        fshape, cshape = ctx.meta
        d = d.contiguous()
        n, fx, fy, fz, c = fshape
        _, cx, cy, cz, _ = cshape
        dc = torch.empty(cshape, dtype=d.dtype, device=d.device)
        call("upsample_add_bwd", _p(d), _p(dc), n, fx, fy, fz, cx, cy, cz, c, _dt(d), 0, _s())
        return d, dc


class SubsampleFn(torch.autograd.Function):
    """Every s-th voxel per axis (the spatial part of a stride-s 1x1x1 convolution)."""

    @staticmethod
    def forward(ctx, x, s):
        _chk(x)
        n, gx, gy, gz, c = x.shape
        o = [(g - 1) // s + 1 for g in (gx, gy, gz)]
        y = torch.empty((n, o[0], o[1], o[2], c), dtype=x.dtype, device=x.device)
        call("subsample3d", _p(x), _p(y), n, gx, gy, gz, c, s, 0, _dt(x), _s())
        ctx.meta = (x.shape, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        shape, s = ctx.meta
        dy = dy.contiguous()
        n, gx, gy, gz, c = shape
        dx = torch.empty(shape, dtype=dy.dtype, device=dy.device)
        call("subsample3d", _p(dy), _p(dx), n, gx, gy, gz, c, s, 1, _dt(dy), _s())
        return dx, None


class AddReluFn(torch.autograd.Function):
    """y = relu(a + b): the residual join of a bottleneck (reference feature_extractor.py:63-66)."""

    @staticmethod
    def forward(ctx, a, b, relu):
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        y = torch.empty_like(a)
        call("add_relu", _p(a), _p(b), _p(y), a.numel(), int(relu), _dt(a), _s())
        ctx.relu = relu
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            g = torch.empty_like(dy)
            call("relu_backward", _p(y), _p(dy), _p(g), dy.numel(), _dt(dy), _s())
        else:
            g = dy
        return g, g, None


# ======================================================================================================================
# Swin-3D pieces (csrc/swin.hip)
# ======================================================================================================================
def patchify(x, patch):
    """[N,X,Y,Z,4] -> [N,X/p,Y/p,Z/p,4p^3] (the network input carries no gradient, so this is not an autograd node)."""
    x = x.contiguous()
    _chk(x)
    n, gx, gy, gz, c = x.shape
    if c != 4:
        raise lib.NrpnError("patchify expects the 4-channel rgb-sigma grid")
    y = torch.empty((n, gx // patch, gy // patch, gz // patch, 4 * patch ** 3), dtype=x.dtype, device=x.device)
    call("patchify", _p(x), _p(y), n, gx, gy, gz, patch, _dt(x), _s())
    return y


def _slots_direct(gs, bs, c):
    """Both parameter gradients of a normalisation layer go to fp32 arena slots of c contiguous elements: the kernels may add in place."""
    return (gs is not None and bs is not None and gs.slot.dtype == torch.float32 and bs.slot.dtype == torch.float32
            and gs.slot.numel() == c and bs.slot.numel() == c and gs.slot.is_contiguous() and bs.slot.is_contiguous())


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(C) on the last dimension of a channels-last token tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        _chk(x)
        c = x.shape[-1]
        rows = x.numel() // c
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        call("layernorm_fwd", _p(x), _p(y), _p(g32), _p(b32), _p(mean), _p(rstd), rows, c, float(eps), _dt(x), _s())
        ctx.save_for_backward(x, g32, mean, rstd)
        ctx.sinks = (_sink(gamma), _sink(beta))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        c = x.shape[-1]
        dx = torch.empty_like(x)
        ws = torch.empty(query("layernorm_workspace_bytes", x.numel() // c, c), dtype=torch.uint8, device=x.device)
        gs, bs = ctx.sinks
        if _slots_direct(gs, bs, c):
            # arena training: the parameter-gradient finish adds straight into the two slots (round 6: ~100 torch adds per Swin-S step gone)
            call("layernorm_bwd", _p(x), _p(dy), _p(dx), _p(g32), _p(mean), _p(rstd), _p(gs.slot), _p(bs.slot), x.numel() // c, c, _dt(x), 1, _p(ws), _s())
            gs.notify()
            bs.notify()
            return dx, None, None, None
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        call("layernorm_bwd", _p(x), _p(dy), _p(dx), _p(g32), _p(mean), _p(rstd), _p(dg), _p(db), x.numel() // c, c, _dt(x), 0, _p(ws), _s())
        if gs is not None:
            gs.slot.add_(dg)
            gs.notify()
            dg = None
        if bs is not None:
            bs.slot.add_(db)
            bs.notify()
            db = None
        return dx, dg, db, None


class GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _chk(x)
        y = torch.empty_like(x)
        call("gelu", _p(x), None, _p(y), x.numel(), 0, _dt(x), _s())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        call("gelu", _p(x), _p(dy), _p(dx), x.numel(), 1, _dt(x), _s())
        return dx


class ScaleAddFn(torch.autograd.Function):
    """y = a + scale[n] * b -- residual join with the StochasticDepth('row') factor (scale None: plain add)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a, b = a.contiguous(), b.contiguous()
        _chk(a, b)
        y = torch.empty_like(a)
        call("scale_add", _p(a), _p(b), _p(scale), _p(y), a.shape[0], a[0].numel(), _dt(a), _s())
        ctx.save_for_backward(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        dy = dy.contiguous()
        if scale is None:
            return dy, dy, None
        db = torch.empty_like(dy)
        call("scale_add", None, _p(dy), _p(scale), _p(db), dy.shape[0], dy[0].numel(), _dt(dy), _s())
        return dy, db, None


class PatchMergeFn(torch.autograd.Function):
    """[N,X,Y,Z,C] -> [N,ceil(X/2),ceil(Y/2),ceil(Z/2),8C], reference PatchMerging gather order (feature_extractor.py:403-419)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        _chk(x)
        n, gx, gy, gz, c = x.shape
        y = torch.empty((n, (gx + 1) // 2, (gy + 1) // 2, (gz + 1) // 2, 8 * c), dtype=x.dtype, device=x.device)
        call("patch_merge", _p(x), _p(y), n, gx, gy, gz, c, 0, _dt(x), _s())
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        n, gx, gy, gz, c = ctx.shape
        dx = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device)
        call("patch_merge", _p(dy), _p(dx), n, gx, gy, gz, c, 1, _dt(dy), _s())
        return dx


class WindowAttnFn(torch.autograd.Function):
    """Shifted-window attention core between the qkv and proj Linears (reference feature_extractor.py:424-530), window
    4x4x4, head_dim 32.  ``qkv_bias`` is what zero-padded tokens turn into after the qkv Linear."""

    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, rel_index, heads, shift):
        qkv = qkv.contiguous()
        _chk(qkv, table, rel_index)
        n, gx, gy, gz, c3 = qkv.shape
        c = c3 // 3
        qb = qkv_bias.detach().float().contiguous() if qkv_bias is not None else None
        t32 = table.detach().float().contiguous()
        out = torch.empty((n, gx, gy, gz, c), dtype=qkv.dtype, device=qkv.device)
        call("window_attn_fwd", _p(qkv), _p(qb), _p(t32), _p(rel_index), _p(out), n, gx, gy, gz, c, heads, int(shift), _dt(qkv), _s())
        ctx.save_for_backward(qkv, qb, t32, rel_index)
        ctx.meta = (heads, int(shift), any(g % 4 for g in (gx, gy, gz)))
        ctx.sinks = (_sink(qkv_bias), _sink(table))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, qb, t32, rel_index = ctx.saved_tensors
        heads, shift, padded = ctx.meta
        dout = dout.contiguous()
        n, gx, gy, gz, c3 = qkv.shape
        c = c3 // 3
        dqkv = torch.empty_like(qkv)
        dpad = torch.empty(c3, dtype=torch.float32, device=qkv.device) if (padded and qb is not None) else None
        ws = torch.empty(query("window_attn_bwd_workspace_bytes", n, gx, gy, gz, heads), dtype=torch.uint8, device=qkv.device)
        bsink, tsink = ctx.sinks
        if (tsink is not None and dpad is None and tsink.slot.dtype == torch.float32 and tsink.slot.is_contiguous()
                and tsink.slot.numel() == t32.numel()):
            # no padded tokens: the relative-position table is this block's only parameter gradient and nothing else writes its slot -- the
            # table reduction adds into the arena itself (main stream, behind the optimiser's gradient clear), no torch add, no side-stream hop
            call("window_attn_bwd", _p(qkv), _p(qb), _p(t32), _p(rel_index), _p(dout), _p(dqkv), _p(tsink.slot), 0, n, gx, gy, gz, c, heads,
                 shift, _dt(qkv), 1, _p(ws), _s())
            tsink.notify()
            return dqkv, None, None, None, None, None
        dtable = torch.empty_like(t32)
        call("window_attn_bwd", _p(qkv), _p(qb), _p(t32), _p(rel_index), _p(dout), _p(dqkv), _p(dtable), _p(dpad), n, gx, gy, gz, c, heads,
             shift, _dt(qkv), 0, _p(ws), _s())
        if tsink is not None and (dpad is None or bsink is not None):
            # arena training: both parameter gradients are added to their slots here instead of travelling through autograd's AccumulateGrad
            # (which runs on the stream it was created on and would fall out of a captured backward, graphs.py).  The qkv bias also receives
            # the bias gradient of the qkv Linear on the weight-gradient stream: same stream, enqueue order = a fixed summation order.
            def deliver():
                tsink.slot.add_(dtable.view_as(tsink.slot))
                tsink.notify()
                if dpad is not None:
                    bsink.slot.add_(dpad.view_as(bsink.slot))
                    bsink.notify()
            _on_wgrad_stream(qkv.device, [dtable] + ([dpad] if dpad is not None else []), True, deliver)
            return dqkv, None, None, None, None, None
        return dqkv, dpad, dtable, None, None, None


# ======================================================================================================================
# FCOS pieces (csrc/fcos.hip)
# ======================================================================================================================
class GroupNormFn(torch.autograd.Function):
    """nn.GroupNorm(groups, C) (+ fused ReLU) on a channels-last [N,X,Y,Z,C] tensor."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        x = x.contiguous()
        _chk(x)
        n, c = x.shape[0], x.shape[-1]
        rows = x[0].numel() // c
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        mean = torch.empty(n * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty(n * groups, dtype=torch.float32, device=x.device)
        ws = torch.empty(query("groupnorm_workspace_bytes", n, c, groups), dtype=torch.uint8, device=x.device)
        y = torch.empty_like(x)
        call("groupnorm_fwd", _p(x), _p(y), _p(g32), _p(b32), _p(mean), _p(rstd), n, rows, c, groups, float(eps), int(relu), _dt(x), _p(ws), _s())
        ctx.save_for_backward(x, y if relu else None, g32, mean, rstd)
        ctx.meta = (groups, relu)
        ctx.sinks = (_sink(gamma), _sink(beta))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, g32, mean, rstd = ctx.saved_tensors
        groups, relu = ctx.meta
        dy = dy.contiguous()
        n, c = x.shape[0], x.shape[-1]
        dx = torch.empty_like(x)
        ws = torch.empty(query("groupnorm_workspace_bytes", n, c, groups), dtype=torch.uint8, device=x.device)
        gs, bs = ctx.sinks
        if _slots_direct(gs, bs, c):
            call("groupnorm_bwd", _p(x), _p(y), _p(dy), _p(dx), _p(g32), _p(mean), _p(rstd), _p(gs.slot), _p(bs.slot), n, x[0].numel() // c, c, groups,
                 int(relu), _dt(x), 1, _p(ws), _s())
            gs.notify()
            bs.notify()
            return dx, None, None, None, None, None
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        call("groupnorm_bwd", _p(x), _p(y), _p(dy), _p(dx), _p(g32), _p(mean), _p(rstd), _p(dg), _p(db), n, x[0].numel() // c, c, groups,
             int(relu), _dt(x), 0, _p(ws), _s())
        if gs is not None:
            gs.slot.add_(dg)
            gs.notify()
            dg = None
        if bs is not None:
            bs.slot.add_(db)
            bs.notify()
            db = None
        return dx, dg, db, None, None, None


class FcosHeadOutFn(torch.autograd.Function):
    """Head epilogue of one pyramid level: fused-GEMM outputs [rows, wrows] f32 -> (logits [rows], reg [rows, D], ctr [rows])
    with Scale / ReLU / stride (reference fcos.py:104-128)."""

    @staticmethod
    def forward(ctx, cls_out, box_out, scale, stride_mul, norm_reg, reg_dim, ctr_on_reg):
        _chk(cls_out, box_out)
        wrows = cls_out.shape[-1]
        rows = cls_out.numel() // wrows
        sc = scale.detach().float().contiguous()
        dev = cls_out.device
        logits = torch.empty(rows, dtype=torch.float32, device=dev)
        reg = torch.empty((rows, reg_dim), dtype=torch.float32, device=dev)
        ctr = torch.empty(rows, dtype=torch.float32, device=dev)
        call("fcos_head_out_f32", _p(cls_out), _p(box_out), wrows, _p(sc), float(stride_mul), int(norm_reg), reg_dim, int(ctr_on_reg), rows,
             _p(logits), _p(reg), _p(ctr), _s())
        ctx.save_for_backward(box_out, sc)
        ctx.meta = (wrows, rows, float(stride_mul), int(norm_reg), reg_dim, int(ctr_on_reg), cls_out.shape)
        ctx.sink = _sink(scale)
        return logits, reg, ctr

    @staticmethod
    def backward(ctx, d_logits, d_reg, d_ctr):
        box_out, sc = ctx.saved_tensors
        wrows, rows, stride_mul, norm_reg, reg_dim, ctr_on_reg, shape = ctx.meta
        d_cls = torch.empty(shape, dtype=torch.float32, device=box_out.device)
        d_box = torch.empty(shape, dtype=torch.float32, device=box_out.device)
        d_scale = torch.zeros(query("fcos_reduce_floats"), dtype=torch.float32, device=box_out.device)     # [0] result, [1] ticket, partials
        dl = d_logits.contiguous() if d_logits is not None else None
        dr = d_reg.contiguous() if d_reg is not None else None
        dc = d_ctr.contiguous() if d_ctr is not None else None
        call("fcos_head_out_bwd_f32", _p(box_out), wrows, _p(sc), stride_mul, norm_reg, reg_dim, ctr_on_reg, rows, _p(dl), _p(dr), _p(dc),
             _p(d_cls), _p(d_box), _p(d_scale), _s())
        if ctx.sink is not None:        # straight into the trainer's arena: a gradient handed back to autograd would be accumulated outside a
            ctx.sink.slot.add_(d_scale[:1].view_as(ctx.sink.slot))      # captured backward (graphs.py) and replays would miss it
            ctx.sink.notify()
            return d_cls, d_box, None, None, None, None, None
        return d_cls, d_box, d_scale[:1], None, None, None, None


class FcosGeometry:
    """Host description of the flattened FCOS location list (level-major, then scene, then voxel)."""

    def __init__(self, n, dims, strides):
        import ctypes
        self.n, self.levels = int(n), len(dims)
        self.dims = [tuple(int(v) for v in d) for d in dims]
        self.strides = [int(s) for s in strides]
        self.counts = [d[0] * d[1] * d[2] for d in self.dims]            # locations per scene per level
        self.total = self.n * sum(self.counts)
        self._dims = (ctypes.c_int32 * (3 * self.levels))(*[v for d in self.dims for v in d])
        self._strides = (ctypes.c_int32 * self.levels)(*self.strides)
        self.segment_offsets = [0]
        for c in self.counts:
            for _ in range(self.n):
                self.segment_offsets.append(self.segment_offsets[-1] + c)

    def args(self):
        import ctypes
        return self.n, self.levels, ctypes.addressof(self._dims), ctypes.addressof(self._strides)

    @staticmethod
    def sizes(ori_sizes):
        import ctypes
        if ori_sizes is None:
            return None, 0
        buf = (ctypes.c_float * (3 * len(ori_sizes)))(*[float(v) for s in ori_sizes for v in s])
        return buf, ctypes.addressof(buf)


def fcos_gt_summary(gt):
    gt = _f32(gt).contiguous()
    _chk(gt)
    out = torch.empty((gt.shape[0], 8), dtype=torch.float32, device=gt.device)
    if gt.shape[0]:
        call("fcos_gt_summary_f32", _p(gt), gt.shape[0], gt.shape[1], _p(out), _s())
    return out


def fcos_targets(geom, targets, ori_sizes, radius, norm_reg, reg_dim, device):
    """labels int8 [total] in {1,0,-1}, reg_targets f32 [total, reg_dim], num_pos int32 [1] (reference loss.py:270-437)."""
    import ctypes
    summ = [fcos_gt_summary(t) for t in targets]
    offs = [0]
    for t in targets:
        offs.append(offs[-1] + t.shape[0])
    summary = torch.cat(summ) if offs[-1] else None
    host = (ctypes.c_int32 * len(offs))(*offs)
    labels = torch.empty(geom.total, dtype=torch.int8, device=device)
    reg_t = torch.empty((geom.total, reg_dim), dtype=torch.float32, device=device)
    npos = torch.empty(1, dtype=torch.int32, device=device)
    keep, sizes = FcosGeometry.sizes(ori_sizes)
    n, levels, dims, strides = geom.args()
    call("fcos_targets_f32", _p(summary), ctypes.addressof(host), n, levels, dims, strides, sizes, float(radius), int(norm_reg), reg_dim,
         _p(labels), _p(reg_t), _p(npos), _s())
    return labels, reg_t, npos


class FocalLossFn(torch.autograd.Function):
    """sum of sigmoid focal loss (alpha, gamma 2) over the locations with label >= 0; the gradient is produced in the same pass."""

    @staticmethod
    def forward(ctx, logits, labels, alpha):
        logits = logits.contiguous()
        _chk(logits, labels)
        out = torch.empty(query("fcos_reduce_floats"), dtype=torch.float32, device=logits.device)      # [0] result, [1] ticket, partials
        grad = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        call("fcos_focal_f32", _p(logits), _p(labels), logits.numel(), float(alpha), _p(out), _p(grad), _s())
        ctx.save_for_backward(grad)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None


def fcos_scores(geom, logits, ctr, ori_sizes, thresh):
    _chk(logits, ctr)
    scores = torch.empty(geom.total, dtype=torch.float32, device=logits.device)
    keep, sizes = FcosGeometry.sizes(ori_sizes)
    n, levels, dims, strides = geom.args()
    call("fcos_scores_f32", _p(logits), _p(ctr), n, levels, dims, strides, sizes, float(thresh), _p(scores), _s())
    return scores


def fcos_decode(geom, idx, score, reg, ori_sizes, reg_dim, min_size):
    """idx / score: [levels*n, k] from segmented_topk -> boxes [levels*n*k, 6|7], sqrt scores (-1 = dropped), levels (f32)."""
    _chk(idx, score, reg)
    count, k = idx.numel(), idx.shape[1]
    w = 7 if reg_dim == 8 else 6
    dev = reg.device
    boxes = torch.empty((count, w), dtype=torch.float32, device=dev)
    out_s = torch.empty(count, dtype=torch.float32, device=dev)
    out_l = torch.empty(count, dtype=torch.float32, device=dev)
    keep, sizes = FcosGeometry.sizes(ori_sizes)
    n, levels, dims, strides = geom.args()
    call("fcos_decode_f32", _p(idx), _p(score), count, k, _p(reg), n, levels, dims, strides, sizes, reg_dim, float(min_size), _p(boxes),
         _p(out_s), _p(out_l), _s())
    return boxes, out_s, out_l


class _ToChannelsLast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        x = x.float().contiguous()
        _chk(x)
        n, c = x.shape[:2]
        out = torch.empty((n, *x.shape[2:], c), dtype=dtype, device=x.device)
        call("ncdhw_to_ndhwc", _p(x), _p(out), n, c, x[0, 0].numel(), _dt(out), _s())
        return out

    @staticmethod
    def backward(ctx, g):
        return to_channels_first(g.contiguous()), None


def to_channels_last(x, dtype):
    """[N,C,X,Y,Z] fp32 -> [N,X,Y,Z,C] dtype (differentiable).  A tensor that is already channels-last in memory (e.g. the output
    of ``ingest_rgbsigma``) passes through as a view."""
    cl = x.permute(0, 2, 3, 4, 1)
    if cl.is_contiguous() and cl.dtype == dtype and not x.requires_grad:
        return cl
    return _ToChannelsLast.apply(x, dtype)


def ingest_rgbsigma(raw, alpha_mode=0, dtype=torch.float32):
    """On-disk layout (W,L,H,4) f32|uint8 on the device -> the reference's logical [4,W,L,H] scene tensor, backed by
    channels-last memory in the compute dtype (reference datasets.py:39-63; alpha_mode 1 = density_to_alpha, 2 = ScanNet's)."""
    raw = raw.contiguous()
    _chk(raw)
    if raw.dim() != 4 or raw.shape[-1] != 4 or raw.dtype not in (torch.float32, torch.uint8):
        raise lib.NrpnError("ingest_rgbsigma expects a (W,L,H,4) float32 or uint8 tensor")
    out = torch.empty(raw.shape, dtype=dtype, device=raw.device)
    call("ingest_rgbsigma", _p(raw), int(raw.dtype == torch.uint8), _p(out), raw.numel() // 4, int(alpha_mode), _dt(out), _s())
    return out.permute(3, 0, 1, 2)


def as_channels_last(x):
    """Logical [N,C,X,Y,Z] (fp32 / bf16) -> channels-last tensor [N,X,Y,Z,C]: a free view when the memory already is channels-last
    (what the HIP backbones return), one conversion kernel otherwise."""
    cl = x.permute(0, 2, 3, 4, 1)
    if cl.is_contiguous():
        return cl
    if x.dtype == torch.float32 or x.dtype == torch.bfloat16:
        return to_channels_last(x.float(), x.dtype)
    raise lib.NrpnError(f"unsupported dtype {x.dtype}")


def roi_align_rotated_3d_fwd(feat_cl, rois, spatial_scale, output_size, sampling_ratio):
    """feat_cl [N,X,Y,Z,C], rois f32 [R,8] -> [R,pw,pl,ph,C] in feat's dtype (reference ROIAlignRotated3D_cuda.cu:78-170)."""
    feat_cl, rois = feat_cl.contiguous(), _f32(rois).contiguous()
    _chk(feat_cl, rois)
    n, x, y, z, c = feat_cl.shape
    pw, pl, ph = output_size
    out = torch.empty((rois.shape[0], pw, pl, ph, c), dtype=feat_cl.dtype, device=feat_cl.device)
    call("roi_align_rotated_3d_fwd", _p(feat_cl), _p(rois), rois.shape[0], n, x, y, z, c, float(spatial_scale), pw, pl, ph, int(sampling_ratio),
         _p(out), _dt(feat_cl), _s())
    return out


def roi_align_rotated_3d_bwd(grad_cl, rois, feat_shape, spatial_scale, output_size, sampling_ratio):
    """grad_cl [R,pw,pl,ph,C] -> gradient of the feature map [N,X,Y,Z,C] (deterministic fixed-point accumulation)."""
    grad_cl, rois = grad_cl.contiguous(), _f32(rois).contiguous()
    _chk(grad_cl, rois)
    n, x, y, z, c = feat_shape
    pw, pl, ph = output_size
    gi = torch.empty(feat_shape, dtype=grad_cl.dtype, device=grad_cl.device)
    ws = torch.empty(query("roi_align_rotated_3d_bwd_workspace_bytes", n, x, y, z, c), dtype=torch.uint8, device=grad_cl.device)
    call("roi_align_rotated_3d_bwd", _p(grad_cl), _p(rois), rois.shape[0], n, x, y, z, c, float(spatial_scale), pw, pl, ph, int(sampling_ratio),
         _p(gi), _p(ws), _dt(grad_cl), _s())
    return gi


def ingest_augment(raw, alpha_mode, dtype, plan):
    """``ingest_rgbsigma`` + the training augmentation of ``plan`` (datasets.AugPlan) in one pass on the device: 90-degree rotation and
    flips as index remaps, rotate_and_scale_scene as a trilinear resample (reference datasets.py:109-163, 291-329)."""
    import ctypes
    raw = raw.contiguous()
    _chk(raw)
    if raw.dim() != 4 or raw.shape[-1] != 4 or raw.dtype not in (torch.float32, torch.uint8):
        raise lib.NrpnError("ingest_augment expects a (W,L,H,4) float32 or uint8 tensor")
    w, l, h = (int(v) for v in raw.shape[:3])
    ow, ol, oh = ((l, w, h) if plan.z_up else (h, l, w)) if plan.rot90 else (w, l, h)
    out = torch.empty((ow, ol, oh, 4), dtype=dtype, device=raw.device)
    xf = plan.xform()
    host, hp = None, 0
    if xf is not None:
        host = (ctypes.c_float * 9)(*[float(v) for v in xf.reshape(-1)])
        hp = ctypes.addressof(host)
    call("ingest_augment", _p(raw), int(raw.dtype == torch.uint8), _p(out), w, l, h, int(alpha_mode), _dt(out), int(plan.rot90), int(plan.z_up),
         int(plan.flips[0]), int(plan.flips[1]), hp, _s())
    return out.permute(3, 0, 1, 2)


def stack_scenes(meshes):
    """torch.stack for scene tensors [4,W,L,H]: keeps channels-last memory when every scene has it (no layout round trip)."""
    if len(meshes) == 1:           # a view: torch.stack would copy the whole scene (65 MB at 160^3)
        return meshes[0].unsqueeze(0)
    if all(m.permute(1, 2, 3, 0).is_contiguous() for m in meshes):
        return torch.stack([m.permute(1, 2, 3, 0) for m in meshes], dim=0).permute(0, 4, 1, 2, 3)
    return torch.stack(meshes, dim=0)


def to_channels_first(x):
    """[N,X,Y,Z,C] -> [N,C,X,Y,Z] fp32 (a real copy; use ``x.permute(0,4,1,2,3)`` for a free view)."""
    _chk(x)
    n, c = x.shape[0], x.shape[-1]
    out = torch.empty((n, c, *x.shape[1:4]), dtype=torch.float32, device=x.device)
    call("ndhwc_to_ncdhw", _p(x), _p(out), n, c, x[0, ..., 0].numel(), _dt(x), _s())
    return out


# ======================================================================================================================
# flat-arena optimiser
# ======================================================================================================================
def grad_sumsq(grad_flat, out, grad_scale=1.0):
    call("grad_sumsq", _p(grad_flat), grad_flat.numel(), float(grad_scale), _p(out), _s())


def adamw_step(p, g, m, v, sumsq, max_norm, lr, betas, eps, wd, step, grad_scale=1.0, shadow=None, zero_grad=False):
    """``zero_grad``: the kernel clears the gradient it has just consumed (no separate fill of the arena)."""
    call("adamw_step_zero_grad" if zero_grad else "adamw_step", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(sumsq), float(grad_scale), float(max_norm),
         float(lr), float(betas[0]), float(betas[1]), float(eps), float(wd), int(step), _p(shadow), _s())


# ======================================================================================================================
# ROIPool without the RoIAlign op (the reference CLI's default second-stage pooling, detector.py:264-438) -- csrc/roipool.hip
# ======================================================================================================================
class RoiPoolFn(torch.autograd.Function):
    """Per-RoI pyramid features of ONE scene: ``levels`` = the scene's maps, logical [C, X, Y, Z] (channels-last memory is consumed as is);
    ``meta`` = i32 [R, 6] crops (kind 'aabb') or f32 [R, 7] enlarged boxes ('pooling' / 'interpolation'); ``level_of`` i32 [R];
    ``scales`` = input voxels per level voxel.  -> f32 [R, C, o0, o1, o2] (a channels-last-backed view).  One launch per level, forward and
    backward; no host synchronisation."""

    @staticmethod
    def forward(ctx, kind, meta, level_of, scales, output_size, *levels):
        o0, o1, o2 = (int(v) for v in output_size)
        R = int(meta.shape[0])
        bad = [f for f in levels if not f.is_cuda or f.dtype not in (torch.float32, torch.bfloat16)]
        if bad:       # say what is supported instead of a generic "not a CUDA tensor" from deeper down
            raise lib.NrpnError(f"ROIPool runs on the HIP kernels only (csrc/roipool.hip): feature maps must be CUDA tensors in float32 or "
                                f"bfloat16, got {bad[0].device} / {bad[0].dtype}; the torch restatement of the reference's pooling "
                                "(oracle/roipool.py) is test infrastructure, not a fallback")
        cls = [as_channels_last(f[None])[0] for f in levels]          # [X, Y, Z, C]
        C = int(cls[0].shape[-1])
        dev = cls[0].device
        out = torch.zeros((R, o0, o1, o2, C), dtype=torch.float32, device=dev)
        arg = torch.full((R, o0, o1, o2, C), -1, dtype=torch.int32, device=dev) if kind != "interpolation" else None
        meta, level_of = meta.contiguous(), level_of.contiguous()
        _chk(meta, level_of, *cls)
        for l, f in enumerate(cls):
            X, Y, Z = (int(v) for v in f.shape[:3])
            if kind == "aabb":
                call("roipool_aabb_fwd", _p(f), X, Y, Z, C, _p(meta), _p(level_of), l, R, o0, o1, o2, _p(out), _p(arg), _dt(f), _s())
            else:
                call("roipool_obb_fwd", _p(f), X, Y, Z, C, _p(meta), _p(level_of), l, R, float(scales[l]), 1 if kind == "interpolation" else 0,
                     o0, o1, o2, _p(out), _p(arg), _dt(f), _s())
        ctx.save_for_backward(meta, level_of, *([arg] if arg is not None else []))
        ctx.info = (kind, [float(v) for v in scales], (o0, o1, o2), [(tuple(f.shape), f.dtype) for f in cls])
        return out.permute(0, 4, 1, 2, 3)

    @staticmethod
    def backward(ctx, g):
        kind, scales, (o0, o1, o2), shapes = ctx.info
        meta, level_of, *rest = ctx.saved_tensors
        arg = rest[0] if rest else None
        R = int(meta.shape[0])
        g = g.permute(0, 2, 3, 4, 1).contiguous().float()
        grads = []
        for l, ((X, Y, Z, C), dt) in enumerate(shapes):
            if not ctx.needs_input_grad[5 + l]:
                grads.append(None)
                continue
            df = torch.empty((X, Y, Z, C), dtype=dt, device=g.device)
            ws = torch.empty(query("roipool_bwd_workspace_bytes", X, Y, Z, C), dtype=torch.uint8, device=g.device)
            code = F32 if dt == torch.float32 else BF16
            if kind == "aabb":
                call("roipool_aabb_bwd", _p(g), _p(arg), _p(meta), _p(level_of), l, R, X, Y, Z, C, o0, o1, o2, _p(df), _p(ws), code, _s())
            else:
                call("roipool_obb_bwd", _p(g), _p(arg), _p(meta), _p(level_of), l, R, scales[l], 1 if kind == "interpolation" else 0, X, Y, Z, C,
                     o0, o1, o2, _p(df), _p(ws), code, _s())
            grads.append(df.permute(3, 0, 1, 2))
        return (None, None, None, None, None, *grads)

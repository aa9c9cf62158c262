"""hipGraph capture of the backbone + FPN ("trunk") of a training step.

A ResNet-50 / Swin-S training step is 500-800 C-ABI crossings and ~70-200 autograd nodes of ~20 us kernels: the host needs 15-24 ms to
enqueue what the GPU runs in 11-17 ms (bench.py ``host``).  The trunk has static shapes and no host decisions -- target assignment and the
sampler's read-back happen before it, the sampled-cone head and the loss (whose launch sizes depend on the draw) after it -- so its forward and
its backward are captured once per input shape as two HIP graphs and replayed with one launch each:

    step:   targets + sampler (eager, own stream)  ->  [graph: backbone + FPN forward]  ->  cone head + loss (eager)
            ->  cone head backward (eager)  ->  [graph: FPN + backbone backward, weight gradients on the side stream inside the graph]
            ->  clip + AdamW (eager)

Capture does not execute anything, so it happens in the middle of ordinary training: the first ``warmup`` calls run eagerly (lazy one-time
work: dynamic-LDS limits, workspaces, tap masks), the next call captures both graphs (the backward one through ``torch.autograd.backward``
on the captured outputs, as ``torch.cuda.make_graphed_callables`` does) and then replays the forward one for real.  Everything a replay
touches lives at fixed addresses: the static input copy, the graph's private memory pool, the trainer's arenas (parameters, gradients, bf16
shadow, dgrad operands) and the BatchNorm buffers.  Once-per-update refreshes that the trunk performs itself (stem / non-arena weight packs)
are forced into the graph by bumping the weight epoch before capture; those the trainer performs eagerly (AdamW's bf16 shadow, the dgrad
operand transposition on the side stream) stay eager and are ordered before the replay by ordinary stream semantics.

Results are bit-identical to the eager path (tests/test_gpu_graph.py).  Opt-in: ``model.use_graph = True`` or NRPN_GRAPH=1."""
import os
import weakref

import torch

from . import ops

ENABLED = [os.environ.get("NRPN_GRAPH", "0") == "1"]
SINK_GENERATION = [0]     # bumped by engine.FlatTrainer.__init__: a new trainer installs new arenas / GradSinks, every capture made before is stale
CAPTURING = [False]       # True while a trunk is being captured (nothing executes): the trainer's gradient notifications are ignored meanwhile.
# NOTE for new trunk ops: every parameter gradient must be ADDED INTO ITS GradSink by the backward kernel sequence itself.  A gradient handed
# back to autograd is accumulated by an AccumulateGrad node on the stream that node was created on -- outside the captured backward -- and
# replays would silently miss it (what ops.WindowAttnFn did before round 4; tests/test_gpu_graph.py holds every backbone to bit-identity).


class _Token:
    __slots__ = ("__weakref__",)


class _Captured:
    __slots__ = ("static_x", "outs", "gouts", "fwd", "bwd", "pool", "dummy", "node", "backward_done", "on_backward")

    def pending(self):
        """True while a replayed forward's saved activations (static, shared by every replay) are still waiting for their backward: the
        autograd node of the last replay is alive and has not run.  A node that was dropped without a backward (a train-mode pass under
        grad mode that never called backward) is garbage by then and does not count."""
        return self.node is not None and self.node() is not None and not self.backward_done


class _GraphedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cap, dummy):
        if cap.pending():
            raise RuntimeError("GraphedBackbone: a second captured forward before the backward of the first (the captured activations are "
                               "static: the first pass' gradients would be computed from the second pass' activations)")
        cap.fwd.replay()
        # the replay rewrote the BatchNorm running statistics behind torch's back (no Python of BatchNormFn.forward runs): the eval-mode
        # (scale, shift) folds cached on this epoch are stale
        ops.BN_STATS_EPOCH[0] += 1
        ctx.cap = cap
        ctx.token = _Token()             # dies with the autograd node: tells a pass that was dropped without backward from one still waiting
        cap.node, cap.backward_done = weakref.ref(ctx.token), False
        return tuple(o.detach() for o in cap.outs)

    @staticmethod
    def backward(ctx, *gs):
        cap = ctx.cap
        if cap.bwd is None:
            # forward-only capture: the backward runs EAGERLY through the autograd graph that was recorded while the forward was captured --
            # its saved activations are the static buffers every replay rewrites -- kept alive across steps with retain_graph.  The kernels,
            # their order and their streams (weight gradients on the side stream) are those of the eager step: only the forward's enqueue
            # cost is gone.  Re-entrant autograd, as torch.utils.checkpoint does it.
            sel = [(o, g) for o, g in zip(cap.outs, gs) if g is not None]
            torch.autograd.backward([o for o, _ in sel], [g for _, g in sel], retain_graph=True)
            cap.backward_done = True
            return None, None
        for sg, g in zip(cap.gouts, gs):
            if g is None:
                sg.zero_()
            else:
                sg.copy_(g)
        ops._wait_dgrad_operands()       # the side stream's operand refresh of the last optimiser step (an eager event) precedes the replay
        cap.bwd.replay()
        cap.backward_done = True
        if cap.on_backward is not None:
            cap.on_backward()            # the trunk's gradients are in the arena now: the trainer may launch their buckets (engine.FlatTrainer)
        return None, None


class GraphedBackbone:
    """``backbone(x)`` through captured forward / backward graphs when training with gradients on a CUDA tensor; eager otherwise."""

    def __init__(self, backbone, warmup=2, max_shapes=4, backward="graph"):
        """backward = "graph": forward and backward captured (one launch each); "eager": only the forward is captured, the backward runs
        eagerly on the recorded autograd graph (for GPU-bound steps whose backward relies on the weight-gradient stream's overlap, which a
        captured backward does not reproduce: VGG19 -- measured 4-6 % slower fully captured)."""
        if backward not in ("graph", "eager"):
            raise ValueError("backward must be 'graph' or 'eager'")
        self.backward_mode = backward
        self.backbone = backbone
        self.warmup = int(warmup)
        self.max_shapes = int(max_shapes)     # every captured input shape pins its own activation pool: scenes of further shapes run eagerly
        self.captured = {}
        self.calls = {}
        self._signature = None
        self.eager_fallbacks = 0              # forwards that ran eagerly because a captured pass was still waiting for its backward

    def _sink_signature(self):
        """What a capture bakes in besides shapes: which parameters deliver gradients and WHERE (the trainer's arena addresses).  A second
        FlatTrainer on the same model (resume, re-bucketing, a new optimiser) or a requires_grad toggle changes it; captures made under
        another signature would keep adding gradients into the old -- possibly freed -- arena."""
        sig = [SINK_GENERATION[0]]
        for p in self.backbone.parameters():
            k = getattr(p, "_nrpn_sink", None)
            sig.append((p.requires_grad, k.slot.data_ptr() if (k is not None and p.requires_grad) else 0))
        return tuple(sig)

    def __call__(self, x):
        bb = self.backbone
        if not (x.is_cuda and bb.training and torch.is_grad_enabled()) or torch.cuda.is_current_stream_capturing():
            return bb(x)
        sig = self._sink_signature()
        if sig != self._signature:           # new trainer / arenas / trainable set: every capture is stale (their pools are released with them)
            self.captured.clear()
            self.calls.clear()
            self._sinks_ok = None
            self._signature = sig
        if not self._sinks_everywhere():
            return bb(x)
        # fp32 and bf16x3 share compute_dtype == float32 but run different kernels (split operands, a captured weight split): a capture made
        # in one arithmetic mode must not be replayed in the other
        key = (tuple(x.shape), x.dtype, bb.compute_dtype, bool(ops.SPLIT3[0]))
        n = self.calls.get(key, 0)
        self.calls[key] = n + 1
        if n < self.warmup:
            return bb(x)
        cap = self.captured.get(key)
        if cap is None:
            if len(self.captured) >= self.max_shapes:
                return bb(x)
            cap = self.captured[key] = self._capture(x)
        if cap.pending():
            # two trunk forwards before one backward (loss(A) + loss(B)): the captured activations are static, so the second pass runs eagerly
            self.eager_fallbacks += 1
            return bb(x)
        if cap.static_x.data_ptr() != x.data_ptr():
            cap.static_x.data.copy_(x)       # (.data: the refill must not bump the version of a tensor the recorded autograd graph saved)
        return _GraphedFn.apply(cap, cap.dummy)

    def _sinks_everywhere(self):
        """A replayed backward delivers gradients only through the fixed addresses the capture saw: every trainable parameter of the trunk
        must accumulate into a trainer's flat arena (ops.GradSink).  Without a FlatTrainer the trunk stays eager."""
        ok = getattr(self, "_sinks_ok", None)
        if ok is None:
            ok = self._sinks_ok = all(getattr(p, "_nrpn_sink", None) is not None for p in self.backbone.parameters() if p.requires_grad)
        return ok

    def _capture(self, x):
        bb = self.backbone
        cap = _Captured()
        cap.node, cap.backward_done, cap.on_backward = None, True, None
        trainer = getattr(next(iter(bb.parameters())), "_nrpn_trainer", None)
        if trainer is not None and trainer() is not None and self.backward_mode == "graph":
            cap.on_backward = trainer().trunk_gradients_ready(list(bb.parameters()))      # (an eager backward notifies the trainer itself)
        cap.static_x = x.detach().clone()
        cap.dummy = torch.zeros(1, device=x.device, requires_grad=True)
        ops.wgrad_stream_join()
        ops.weights_changed()            # once-per-update packs inside the trunk (stem, non-arena weights) become part of the graph
        torch.cuda.synchronize()
        cap.pool = torch.cuda.graph_pool_handle()
        cap.fwd, cap.bwd = torch.cuda.CUDAGraph(), (torch.cuda.CUDAGraph() if self.backward_mode == "graph" else None)
        CAPTURING[0] = True
        try:
            with torch.cuda.graph(cap.fwd, pool=cap.pool):
                with torch.enable_grad():
                    outs = tuple(bb(cap.static_x))
            cap.outs = outs
            cap.gouts = tuple(torch.zeros_like(o) for o in outs)
            live = [o for o in outs if o.requires_grad]
            if len(live) != len(outs):
                raise RuntimeError("GraphedBackbone: every trunk output must require a gradient (is the backbone frozen?)")
            if cap.bwd is not None:
                with torch.cuda.graph(cap.bwd, pool=cap.pool):
                    torch.autograd.backward(outs, cap.gouts)
        finally:
            CAPTURING[0] = False
        torch.cuda.synchronize()
        return cap

"""GPU: engine.FlatTrainer (flat arenas, gradient sinks, fused clip + AdamW through raw pointers) against the same HIP model
trained with torch.optim.AdamW + clip_grad_norm_ (reference semantics: run_rpn.py:345-349, 388-395)."""
import numpy as np
import pytest
import torch

from test_gpu_e2e import T, build, scene

pytestmark = pytest.mark.gpu


def _batch(golden, dev):
    g = golden("train_obb")
    xs = [scene(s, 200 + i).to(dev) for i, s in enumerate(g["shapes"])]
    gts = [T(g[f"gt{i}"], dev) for i in range(len(xs))]
    return xs, gts, T(g["pos_idx"], dev), T(g["neg_idx"], dev)


def _loss(model, xs, gts, pos, neg):
    model.rpn.sampler_hook = lambda labels: (pos, neg)
    _, losses, _ = model(xs, gts)
    return losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]


def test_two_runs_give_bit_identical_gradients(golden, dev):
    """Every reduction on the training path is ordered (K-slice partials, bias partials, BN slab partials, no fp32 atomics): two
    forward/backward runs of the same model on the same batch agree bit for bit."""
    xs, gts, pos, neg = _batch(golden, dev)
    m = build(True, 160, dev).train()
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        _loss(m, xs, gts, pos, neg).backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]))
    assert torch.equal(grads[0], grads[1]), (grads[0] - grads[1]).abs().max().item()


def test_flat_trainer_arena_equals_autograd_grads(golden, dev):
    """Backward kernels accumulate straight into the trainer's flat gradient arena (ops.GradSink); the arena must EQUAL what plain
    autograd accumulates into p.grad, on the first step (counts being learned) and on the second (buckets launched by count)."""
    from nerf_rpn_amd.engine import FlatTrainer
    xs, gts, pos, neg = _batch(golden, dev)
    ref = build(True, 160, dev).train()
    _loss(ref, xs, gts, pos, neg).backward()
    plain = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    m = build(True, 160, dev).train()
    tr = FlatTrainer(m, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1)
    for _ in range(2):
        tr.g_arena.zero_()
        _loss(m, xs, gts, pos, neg).backward()
        tr.sync_gradients()
        got = tr.flat_grads()
        assert torch.equal(got, plain), (got - plain).abs().max().item() / plain.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_flat_trainer_trains_like_torch_adamw(dtype, golden, dev):
    """Three optimiser steps.  The GEMM-layout weight copies must follow the raw-pointer AdamW update (round-1 bug: they were
    cached on tensor._version and every conv kept the step-0 weights): the loss trajectory and the weights must track the same
    model trained by torch.optim.AdamW, and every packed module must repack exactly once per step."""
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.engine import FlatTrainer
    xs, gts, pos, neg = _batch(golden, dev)
    lr, steps = 3e-4, 3

    ref = build(True, 160, dev).train()
    ref.set_compute_dtype(dtype)
    opt = torch.optim.AdamW(ref.parameters(), lr=lr, weight_decay=0.01)
    # conv biases that feed a train-mode BatchNorm have an exactly-zero gradient in exact arithmetic: what both implementations see is
    # rounding noise, which Adam turns into +-lr steps of arbitrary sign -- those tensors are excluded from the weight comparison
    mods = dict(ref.named_modules())
    dead = {name for name, _ in ref.named_parameters()
            if name.startswith("backbone.layers") and name.endswith(".bias") and isinstance(mods[name.rsplit(".", 1)[0]], torch.nn.Conv3d)}
    assert len(dead) == 17
    ref_losses, sig, ref_grads = [], None, []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = _loss(ref, xs, gts, pos, neg)
        loss.backward()
        ref_grads.append([p.grad.detach().clone() for p in ref.parameters()])
        # entries whose gradient is well above their tensor's rounding noise on EVERY step (see the weight comparison below)
        step_sig = torch.cat([((p.grad.abs() > 1e-2 * p.grad.abs().max()) & (name not in dead)).reshape(-1) for name, p in ref.named_parameters()])
        sig = step_sig if sig is None else (sig & step_sig)
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.1)
        opt.step()
        ref_losses.append(loss.item())
    with torch.no_grad():
        ref_final = _loss(ref, xs, gts, pos, neg).item()

    m = build(True, 160, dev).train()
    m.set_compute_dtype(dtype)
    tr = FlatTrainer(m, lr=lr, weight_decay=0.01, clip_grad_norm=0.1)
    before = dict(ops.PACK_COUNT)
    losses, flipped = [], {}
    names = [n for n, _ in m.named_parameters()]
    for it in range(steps):
        loss = _loss(m, xs, gts, pos, neg)
        loss.backward()
        torch.cuda.synchronize()
        # routing-flip detector (tools/diag_trainer_cone.py): per tensor, the trainer's gradient against autograd's on the SAME step.  After
        # an identical first step the weights differ in the last bit; train-mode BatchNorm at batch 1 amplifies that to 2e-3 .. 7e-3 of every
        # tensor's gradient scale on steps 2 and 3 (measured, round 5) -- and a ReLU / max-pool input within rounding distance of zero that
        # changes side adds percents to the tensors downstream of it (measured: layers.5.6 / 5.7, fpn_convs.1: 1.3e-2 .. 2.5e-2)
        for k, (p, gr) in enumerate(zip(m.parameters(), ref_grads[it])):
            rel = (p.grad - gr).abs().max().item() / (gr.abs().max().item() + 1e-30)
            if rel > 1e-2 and names[k] not in dead:
                flipped[names[k]] = max(flipped.get(names[k], 0.0), rel)
        tr.step()
        losses.append(loss.item())
    # single-weight GEMMs read the arena (fp32 master / bf16 shadow written by AdamW) directly: only the fused cls+bbox head GEMM and
    # the 7^3 stem still repack, exactly once per step; the dgrad operands of all arena weights are refreshed by ONE launch per step
    assert ops.PACK_COUNT["conv"] - before["conv"] == steps, (ops.PACK_COUNT, before)
    assert ops.PACK_COUNT["stem"] - before["stem"] == steps
    # (the first one lazily in the first forward, then one on the side stream right after every optimiser step: ArenaWeights.prefetch_dgrad)
    assert tr.weights.launches["transpose"] == steps + 1 and tr.weights.launches["cast"] == (1 if dtype == torch.bfloat16 else 0), tr.weights.launches
    assert len(tr.weights.entries) >= 28
    with torch.no_grad():
        final = _loss(m, xs, gts, pos, neg).item()
    # the optimiser must visibly move the loss (a forward on stale weights would reproduce losses[0] exactly) ...
    assert max(abs(v - ref_losses[0]) for v in ref_losses[1:] + [ref_final]) > 0.1 * abs(ref_losses[0]), (ref_losses, ref_final)
    # ... and both trainers must follow the same trajectory.  fp32: identical gradients (deterministic kernels), the only difference
    # is the rounding of the AdamW formula; bf16 adds re-rounding of the updated weights.
    tol = 2e-3 if dtype == torch.float32 else 5e-2
    for a, b in zip(losses + [final], ref_losses + [ref_final]):
        assert abs(a - b) <= tol * max(1.0, abs(b)), (losses, final, ref_losses, ref_final)
    after_ref = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    # Adam normalises every entry to a +-lr-sized step, so entries whose gradient is rounding noise (e.g. conv biases in front of
    # BatchNorm: exact gradient 0) step in directions that depend on the last bit of the clip coefficient; the weights are compared
    # where the gradient was significant (> 1 % of its tensor's maximum) on all three steps.
    assert sig.float().mean().item() > 0.02, sig.float().mean().item()
    if dtype == torch.float32:
        # Identical gradients on the first step (bit for bit: test_flat_trainer_arena_equals_autograd_grads); afterwards the two AdamW
        # implementations differ in the last bit of the weights (1.2e-7), which moves a ReLU input that sits within rounding distance of zero
        # across it now and then -- ONE such routing flip at a coarse level changes a handful of gradient entries by percents (tools/
        # diag_trainer_cone.py: step 2, rpn.head.conv.0 + fpn_convs.1 only) and Adam turns that into a fraction of an lr-sized step.
        # Required: 99.9 % of the significant entries within 5 % of the total step budget, none further off than ONE reversed Adam step
        # (an entry whose gradient changed sign on one of the steps travels +lr instead of -lr: 2 lr; measured worst 1.25 lr), and the
        # update as a whole the same vector (cosine > 0.9999, norm within 0.1 %).  The AdamW kernel itself is held to torch.optim.AdamW on
        # IDENTICAL gradients in test_gpu_conv.py::test_layout_roundtrip_and_adamw (1e-6).
        d = (tr.flat_params() - after_ref)[sig].abs()
        assert (d < 0.05 * lr * steps).float().mean().item() > 0.999, (d < 0.05 * lr * steps).float().mean().item()
        # an entry further off than 0.3 lr x steps must BELONG to a tensor whose gradient the detector above saw change by
        # > 1e-2 on some step (a routing flip upstream of it); every other tensor stays inside 0.3 lr x steps (measured on the round-5 tree:
        # NO tensor leaves it -- the allowance below exists for the flipped ones only).  Flipped tensors: at most one
        # reversed Adam step (2 lr; measured 1.25 lr), they are few, and the first step -- identical weights -- has none.
        full = (tr.flat_params() - after_ref).abs()
        off, outside = 0, {}
        for name, p in m.named_parameters():
            n_ = p.numel()
            dd = full[off:off + n_][sig[off:off + n_]]
            off += n_
            if dd.numel() and dd.max().item() >= 0.3 * lr * steps:
                outside[name] = dd.max().item()
        print(f"[trainer] tensors with a detected routing flip: { {k: f'{v:.1e}' for k, v in flipped.items()} }; beyond 0.3 lr x steps: "
              f"{ {k: f'{v / lr:.2f} lr' for k, v in outside.items()} }")
        assert set(outside) <= set(flipped), (outside, flipped)
        assert len(flipped) <= 8, flipped          # a flip is local: the tensors right downstream of one voxel (measured: 4 tensors)
        assert d.max().item() < 2.2 * lr, d.max().item()
        init = torch.cat([p.detach().reshape(-1) for p in build(True, 160, dev).parameters()])
        a, b = (tr.flat_params() - init)[sig].double(), (after_ref - init)[sig].double()
        cos = (a @ b / (a.norm() * b.norm())).item()
        assert cos > 0.9999 and abs((a.norm() / b.norm()).item() - 1.0) < 1e-3, (cos, (a.norm() / b.norm()).item())
    else:
        # bf16 re-rounds the weights every step: activations move by ~2^-9 relative, which flips the sign of individual Adam steps even
        # where the fp32 gradient is comfortably non-zero; the UPDATE as a whole must still point the same way and have the same size
        init = torch.cat([p.detach().reshape(-1) for p in build(True, 160, dev).parameters()])
        a, b = (tr.flat_params() - init)[sig].double(), (after_ref - init)[sig].double()
        cos = (a @ b / (a.norm() * b.norm())).item()
        assert cos > 0.9 and 0.9 < (a.norm() / b.norm()).item() < 1.1, (cos, (a.norm() / b.norm()).item())


def test_wgrad_side_stream_gives_the_same_arena(golden, dev):
    """ConvFn.backward enqueues the weight-gradient kernels of arena training on a second HIP stream (ops.set_wgrad_stream): the gradient
    arena and three optimiser steps must be bit-identical to the single-stream run."""
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.engine import FlatTrainer
    xs, gts, pos, neg = _batch(golden, dev)
    out = {}
    try:
        for side in (False, True):
            ops.set_wgrad_stream(side)
            torch.manual_seed(0)
            m = build(True, 160, dev).train()
            tr = FlatTrainer(m, lr=3e-4, weight_decay=0.01, clip_grad_norm=0.1)
            _loss(m, xs, gts, pos, neg).backward()
            assert ops._WGRAD_SIDE["dirty"] is False            # the end-of-backward callback joined the side stream
            g = tr.g_arena.clone()
            for _ in range(3):
                tr.step()
                _loss(m, xs, gts, pos, neg).backward()
            out[side] = (g, tr.p_arena.clone(), bool(ops._WGRAD_SIDE["streams"]))
    finally:
        ops.set_wgrad_stream(True)
    assert out[True][2], "the side stream was never created"
    assert torch.equal(out[False][0], out[True][0]), (out[False][0] - out[True][0]).abs().max().item()
    assert torch.equal(out[False][1], out[True][1]), (out[False][1] - out[True][1]).abs().max().item()

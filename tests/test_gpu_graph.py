"""graphs.GraphedBackbone: backbone + FPN forward / backward of a training step as captured HIP graphs.  The replayed step must be the eager
step bit for bit: same losses, same gradient arena, same weights after the optimiser, over several steps that include the two eager warm-up
steps, the capturing step and replays -- with scenes that change from step to step (the static input copy) and with the weight-gradient
side stream inside the captured backward."""
import pytest
import torch

from test_gpu_e2e import T, build, scene

pytestmark = pytest.mark.gpu


def _run(dev, golden, use_graph, backbone, dtype, steps=6):
    from nerf_rpn_amd.engine import FlatTrainer
    g = golden("train_obb")
    shape = tuple(int(v) for v in g["shapes"][0])
    gts = [T(g["gt0"], dev)]
    pos, neg = T(g["pos_idx"], dev), T(g["neg_idx"], dev)
    m = build(True, 160, dev, backbone=backbone, sd=0.0).train()
    m.set_compute_dtype(dtype)
    m.use_graph = use_graph
    m.rpn.sampler_hook = lambda labels: (pos, neg)
    tr = FlatTrainer(m, lr=3e-4, weight_decay=0.01, clip_grad_norm=0.1)
    losses, arenas = [], []
    for it in range(steps):
        x = scene(shape, 200 + (it % 3)).to(dev)          # three different grids in turn: the captured input buffer is refilled every step
        _, l, _ = m([x], gts)
        (l["loss_objectness"] + 5.0 * l["loss_rpn_box_reg"]).backward()
        torch.cuda.synchronize()
        arenas.append(tr.g_arena.clone())
        tr.step()
        losses.append((l["loss_objectness"].item(), l["loss_rpn_box_reg"].item()))
    captured = len(m._trunk.captured) if m._trunk is not None else 0
    return losses, arenas, tr.flat_params().clone(), {k: v.clone() for k, v in m.backbone.state_dict().items() if "running" in k or "tracked" in k}, captured


@pytest.mark.parametrize("backbone,dtype", [("vgg", torch.float32), ("vgg", torch.bfloat16), ("resnet", torch.bfloat16), ("swin", torch.float32),
                                            ("swin", torch.bfloat16)])
def test_graphed_trunk_is_bit_identical_to_the_eager_step(backbone, dtype, golden, dev):
    eager = _run(dev, golden, False, backbone, dtype)
    graph = _run(dev, golden, True, backbone, dtype)
    assert eager[4] == 0 and graph[4] == 1                    # one capture (one input shape), after the two eager warm-up calls
    assert eager[0] == graph[0], (eager[0], graph[0])
    for i, (a, b) in enumerate(zip(eager[1], graph[1])):
        assert torch.equal(a, b), (i, (a - b).abs().max().item())
    assert torch.equal(eager[2], graph[2])
    for k, v in eager[3].items():
        assert torch.equal(v, graph[3][k]), k


@pytest.mark.parametrize("backbone", ["swin", "vgg"])
def test_graphed_trunk_under_the_fcos_head(backbone, dev):
    """FCOSOverNeRF shares the trunk: same check (losses, arena, weights) with the dense FCOS head behind the captured backbone."""
    import test_gpu_fcos as TF
    from nerf_rpn_amd.engine import FlatTrainer
    shape = (64, 56, 48)
    gt = torch.tensor([[20., 18., 16., 14., 12., 10., 0.3], [40., 30., 28., 16., 18., 12., -0.5], [30., 40., 20., 10., 10., 14., 0.9]], device=dev)
    out = {}
    for use_graph in (False, True):
        m = TF.build(True, backbone, dev).train()
        m.set_compute_dtype(torch.bfloat16)
        m.use_graph = use_graph
        tr = FlatTrainer(m, lr=3e-4, weight_decay=0.01, clip_grad_norm=0.1)
        losses, arenas = [], []
        for it in range(5):
            x = TF.scene(shape, 500 + (it % 2)).to(dev)
            _, l, _ = m([x], [gt])
            (l["loss_cls"] + l["loss_reg"] + l["loss_centerness"]).backward()
            torch.cuda.synchronize()
            arenas.append(tr.g_arena.clone())
            tr.step()
            losses.append(tuple(v.item() for v in l.values()))
        out[use_graph] = (losses, arenas, tr.flat_params().clone(), 0 if m._trunk is None else len(m._trunk.captured))
    assert out[False][3] == 0 and out[True][3] == 1
    assert out[False][0] == out[True][0], (out[False][0], out[True][0])
    for i, (a, b) in enumerate(zip(out[False][1], out[True][1])):
        assert torch.equal(a, b), (i, (a - b).abs().max().item())
    assert torch.equal(out[False][2], out[True][2])


def test_swin_steps_are_bit_reproducible_with_the_weight_gradient_stream(golden, dev):
    """Regression (round 4): a residual join hands the SAME gradient tensor to both branches; autograd then accumulates into it in place on the
    main stream while a weight-gradient kernel on the side stream still reads it, unless something else holds a reference
    (ops._WGRAD_SIDE['keep']).  Two eager runs of the Swin-S model must agree bit for bit -- they did not (gradient arena off by 5e-2)."""
    a = _run(dev, golden, False, "swin", torch.float32, steps=3)
    b = _run(dev, golden, False, "swin", torch.float32, steps=3)
    assert a[0] == b[0]
    for x, y in zip(a[1], b[1]):
        assert torch.equal(x, y), (x - y).abs().max().item()


def test_a_second_trainer_invalidates_the_captures_and_a_second_forward_runs_eagerly(golden, dev):
    """a capture bakes in the FlatTrainer's arena addresses.  (1) A second trainer on the same model (resume, a new
    optimiser) must drop the captures: its gradient arena has to receive the trunk's gradients -- equal, bit for bit, to an eager run under
    the same second trainer.  (2) Two trunk forwards before one backward (loss(A) + loss(B)): the second pass must not overwrite the first
    pass' captured activations -- it runs eagerly, and the summed gradients equal the all-eager ones."""
    from nerf_rpn_amd.engine import FlatTrainer
    g = golden("train_obb")
    shape = tuple(int(v) for v in g["shapes"][0])
    gts = [T(g["gt0"], dev)]
    pos, neg = T(g["pos_idx"], dev), T(g["neg_idx"], dev)

    def run(use_graph):
        m = build(True, 160, dev, backbone="vgg", sd=0.0).train()
        m.set_compute_dtype(torch.bfloat16)
        m.use_graph = use_graph
        m.rpn.sampler_hook = lambda labels: (pos, neg)
        tr = FlatTrainer(m, lr=3e-4, weight_decay=0.01, clip_grad_norm=0.1)

        def one(tr_, seed):
            _, l, _ = m([scene(shape, seed).to(dev)], gts)
            (l["loss_objectness"] + 5.0 * l["loss_rpn_box_reg"]).backward()
            torch.cuda.synchronize()
            ga = tr_.g_arena.clone()
            tr_.step()
            return ga
        for it in range(4):                       # 2 eager warm-ups, the capture, a replay
            one(tr, 200 + it)
        captured_before = 0 if m._trunk is None else len(m._trunk.captured)
        tr2 = FlatTrainer(m, lr=3e-4, weight_decay=0.01, clip_grad_norm=0.1)      # new arenas, new sinks
        arenas = [one(tr2, 300 + it) for it in range(4)]
        old_arena_touched = float(tr.g_arena.abs().max())          # the first trainer's arena (AdamW cleared it) must stay untouched
        # two forwards, one backward
        _, la, _ = m([scene(shape, 400).to(dev)], gts)
        _, lb, _ = m([scene(shape, 401).to(dev)], gts)
        (la["loss_objectness"] + 5.0 * la["loss_rpn_box_reg"] + lb["loss_objectness"] + 5.0 * lb["loss_rpn_box_reg"]).backward()
        torch.cuda.synchronize()
        both = tr2.g_arena.clone()
        fallbacks = 0 if m._trunk is None else m._trunk.eager_fallbacks
        return captured_before, arenas, old_arena_touched, both, fallbacks, (0 if m._trunk is None else len(m._trunk.captured))

    eager, graph = run(False), run(True)
    assert graph[0] == 1 and graph[5] == 1            # captured under the first trainer, re-captured (once) under the second
    assert graph[2] == 0.0, "a replay under the second trainer wrote into the first trainer's gradient arena"
    for i, (a, b) in enumerate(zip(eager[1], graph[1])):
        assert float(b.abs().max()) > 0 and torch.equal(a, b), (i, (a - b).abs().max().item())
    assert graph[4] == 1                              # the second forward of the pair ran eagerly
    assert torch.equal(eager[3], graph[3]), (eager[3] - graph[3]).abs().max().item()


@pytest.mark.parametrize("backbone,dtype", [("vgg", torch.bfloat16), ("vgg", torch.float32), ("resnet", torch.bfloat16)])
def test_forward_only_capture_with_eager_backward_is_bit_identical(backbone, dtype, golden, dev):
    """use_graph = "fwd" (round 5): only the trunk's forward is captured; its backward runs eagerly, every step, on the autograd graph recorded
    during the capture (retain_graph; saved activations = the static buffers each replay rewrites) -- the eager step's kernels, order and
    streams (weight gradients on the side stream), without the forward's enqueue cost.  Losses, the gradient arena of every step, the weights
    and the BatchNorm buffers equal the eager run bit for bit over 6 steps with three different scenes in turn."""
    eager = _run(dev, golden, False, backbone, dtype)
    fwd = _run(dev, golden, "fwd", backbone, dtype)
    assert eager[4] == 0 and fwd[4] == 1
    assert eager[0] == fwd[0], (eager[0], fwd[0])
    for i, (a, b) in enumerate(zip(eager[1], fwd[1])):
        assert torch.equal(a, b), (i, (a - b).abs().max().item())
    assert torch.equal(eager[2], fwd[2])
    for k, v in eager[3].items():
        assert torch.equal(v, fwd[3][k]), k

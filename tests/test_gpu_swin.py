"""GPU parity of the Swin-3D kernels (csrc/swin.hip) against torch fp32 on the SAME inputs, forward and backward, and of
whole blocks / the whole backbone against the oracle restatement (oracle/nets.py, pinned to the reference's
feature_extractor.py:382-789 by tests/golden/make_golden.py)."""
import pytest
import torch
import torch.nn.functional as F

from fixture_init import seeded_state

pytestmark = pytest.mark.gpu


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / (b.double().abs().max() + 1e-12)).item()


def test_layernorm_gelu_merge_patchify(dev):
    from nerf_rpn_amd import ops
    g = torch.Generator().manual_seed(0)
    for c in (96, 768, 3072):
        x = (torch.randn(2, 5, 3, 7, c, generator=g) * 2 + 0.5).to(dev).requires_grad_()
        w = (torch.rand(c, generator=g) + 0.5).to(dev).requires_grad_()
        b = torch.randn(c, generator=g).to(dev).requires_grad_()
        dy = torch.randn(2, 5, 3, 7, c, generator=g).to(dev)
        y = ops.LayerNormFn.apply(x, w, b, 1e-5)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), dy)
        yr = F.layer_norm(x, (c,), w, b, 1e-5)
        rx, rw, rb = torch.autograd.grad(yr, (x, w, b), dy)
        assert rel(y, yr) < 2e-6 and rel(gx, rx) < 1e-5 and rel(gw, rw) < 1e-5 and rel(gb, rb) < 1e-5, c
    x = (torch.randn(3, 4, 5, 6, 96, generator=g) * 3).to(dev).requires_grad_()
    dy = torch.randn(3, 4, 5, 6, 96, generator=g).to(dev)
    y = ops.GeluFn.apply(x)
    (gx,) = torch.autograd.grad(y, x, dy)
    yr = F.gelu(x)
    (rx,) = torch.autograd.grad(yr, x, dy)
    assert rel(y, yr) < 2e-6 and rel(gx, rx) < 2e-6
    # residual join with the per-sample stochastic-depth factor
    a, bb = torch.randn(3, 4, 5, 6, 96, generator=g).to(dev).requires_grad_(), torch.randn(3, 4, 5, 6, 96, generator=g).to(dev).requires_grad_()
    sc = torch.tensor([0.0, 1.25, 1.25], device=dev)
    y = ops.ScaleAddFn.apply(a, bb, sc)
    ga, gb = torch.autograd.grad(y, (a, bb), dy)
    assert rel(y, a + sc.view(3, 1, 1, 1, 1) * bb) < 1e-6 and torch.equal(ga, dy) and torch.equal(gb, sc.view(3, 1, 1, 1, 1) * dy)   # fused multiply-add
    # patch merging gather, odd sizes (reference order x0..x7, feature_extractor.py:669-681)
    x = torch.randn(2, 5, 4, 3, 8, generator=g).to(dev).requires_grad_()
    y = ops.PatchMergeFn.apply(x)
    xp = F.pad(x, (0, 0, 0, 1, 0, 0, 0, 1))
    yr = torch.cat([xp[:, i::2, j::2, k::2] for k in (0, 1) for j in (0, 1) for i in (0, 1)], -1)
    dy = torch.randn_like(yr)
    assert torch.equal(y, yr)
    assert torch.equal(torch.autograd.grad(y, x, dy)[0], torch.autograd.grad(yr, x, dy)[0])
    # patch embedding = gather + 1x1x1 GEMM
    from nerf_rpn_amd.model import hip_nn
    conv = torch.nn.Conv3d(4, 96, 4, stride=4).to(dev)
    xin = torch.rand(2, 4, 21, 16, 12, generator=g).to(dev)
    xcl = xin.permute(0, 2, 3, 4, 1).contiguous()
    y = ops.ConvFn.apply(ops.patchify(xcl, 4), hip_nn._pack_of(conv), 96, False, False, 1, conv.weight, conv.bias)
    yr = conv(xin).permute(0, 2, 3, 4, 1)
    assert rel(y, yr) < 1e-5
    dy = torch.randn_like(yr)
    gw, gb = torch.autograd.grad(y, (conv.weight, conv.bias), dy)
    rw, rb = torch.autograd.grad(yr, (conv.weight, conv.bias), dy)
    assert rel(gw, rw) < 2e-5 and rel(gb, rb) < 2e-5


@pytest.mark.parametrize("shape,heads,shift", [((2, 8, 8, 8), 3, 0), ((2, 8, 8, 8), 3, 2), ((1, 10, 7, 6), 6, 2), ((1, 10, 7, 6), 6, 0),
                                               ((2, 5, 4, 3), 12, 2), ((1, 3, 2, 2), 24, 2), ((1, 12, 9, 4), 3, 2)])
def test_window_attention_matches_oracle(shape, heads, shift, dev):
    """qkv Linear -> attention core -> proj Linear against oracle.nets.window_attention on the same tokens: padding to the
    window, cyclic shift, partly-shifted grids (an axis one window long), -100 region mask, relative position bias."""
    from nerf_rpn_amd.model.feature_extractor import ShiftedWindowAttention
    from oracle import nets as ON
    b, h, w, d = shape
    c = 32 * heads
    att = ShiftedWindowAttention(c, [4, 4, 4], [shift] * 3, heads)
    seeded_state(att, 3)
    orc = ON.WindowAttention(c, heads, shift)
    orc.load_state_dict(att.state_dict())
    att = att.to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, h, w, d, c, generator=g)
    dy = torch.randn(b, h, w, d, c, generator=g)
    xo = x.clone().requires_grad_()
    yo = orc(xo)
    go = torch.autograd.grad(yo, [xo] + list(orc.parameters()), dy)
    xg = x.to(dev).requires_grad_()
    yg = att(xg)
    gg = torch.autograd.grad(yg, [xg] + list(att.parameters()), dy.to(dev))
    assert rel(yg, yo) < 1e-5, rel(yg, yo)
    names = ["x"] + [n for n, _ in orc.named_parameters()]
    assert [n for n, _ in att.named_parameters()] == names[1:]
    for n, a, r in zip(names, gg, go):
        assert rel(a, r) < 5e-5, (n, rel(a, r))


def test_swin_backbone_forward_backward_matches_oracle(dev):
    """swin_t-shaped backbone (depths 2,2,2,2) + FPN on identical inputs vs the oracle: features and every parameter gradient."""
    from nerf_rpn_amd.model.feature_extractor import SwinTransformer_FPN
    from oracle import nets as ON
    bb = SwinTransformer_FPN([4, 4, 4], 96, [2, 2, 2, 2], [3, 6, 12, 24], [4, 4, 4], stochastic_depth_prob=0.0)
    seeded_state(bb, 11)
    orc = ON.SwinFPN(96, (2, 2, 2, 2), (3, 6, 12, 24), 0.0)
    orc.load_state_dict(bb.state_dict())
    bb = bb.to(dev).train()
    orc.train()
    x = torch.rand(2, 4, 44, 36, 20, generator=torch.Generator().manual_seed(1))
    fo = orc(x)
    fg = bb(x.to(dev))
    gen = torch.Generator().manual_seed(2)
    dys = [torch.randn(f.shape, generator=gen) * (torch.rand(f.shape, generator=gen) < 0.05) for f in fo]
    for a, r in zip(fg, fo):
        assert a.shape == r.shape and rel(a, r) < 2e-5, rel(a, r)
    go = torch.autograd.grad(fo, list(orc.parameters()), dys)
    gg = torch.autograd.grad(fg, list(bb.parameters()), [t.to(dev) for t in dys])
    for (n, _), a, r in zip(orc.named_parameters(), gg, go):
        assert rel(a, r) < 2e-4, (n, rel(a, r))


def test_stochastic_depth_rows(dev):
    """train-mode StochasticDepth('row'): every sample's branch is either dropped or scaled by 1/(1-p); eval is the identity."""
    from nerf_rpn_amd.model.feature_extractor import SwinTransformerBlock
    blk = SwinTransformerBlock(96, 3, [4, 4, 4], [0, 0, 0], stochastic_depth_prob=0.5).to(dev)
    x = torch.randn(16, 4, 4, 4, 96, device=dev)
    torch.manual_seed(0)
    blk.eval()
    with torch.no_grad():
        base = blk(x)
        blk.train()
        sc = blk.stochastic_depth.scale(x)
        assert set(sc.tolist()) <= {0.0, 2.0} and 0 < (sc == 0).sum() < 16
        blk.stochastic_depth.scale = lambda t: torch.zeros(t.shape[0], device=dev)
        assert torch.equal(blk(x), x)
        blk.stochastic_depth.scale = lambda t: None
        assert torch.equal(blk(x), base)


def test_swin_bf16_close_to_fp32(dev):
    from nerf_rpn_amd.model.feature_extractor import SwinTransformer_FPN
    bb = SwinTransformer_FPN([4, 4, 4], 96, [2, 2, 2, 2], [3, 6, 12, 24], [4, 4, 4], stochastic_depth_prob=0.0)
    seeded_state(bb, 11)
    bb = bb.to(dev).eval()
    x = torch.rand(1, 4, 48, 48, 32, device=dev)
    with torch.no_grad():
        f32 = bb(x)
        bb.compute_dtype = torch.bfloat16
        f16 = bb(x)
    for a, b in zip(f16, f32):
        assert a.dtype == torch.bfloat16 and rel(a.float(), b) < 0.06, rel(a.float(), b)


@pytest.mark.parametrize("shape,heads,shift", [((2, 8, 8, 8), 3, 0), ((1, 10, 7, 6), 6, 2), ((2, 5, 4, 3), 12, 2), ((1, 12, 9, 4), 3, 2)])
def test_bf16_mfma_attention_matches_valu_kernels(shape, heads, shift, dev):
    """bf16 tensors run the attention on the matrix cores (transposed-orientation MFMA kernels); same bf16 inputs through the
    VALU kernels (fp32 math, checked against the oracle above) must agree up to bf16 rounding of the probabilities, forward
    and backward (dq/dk/dv through the qkv Linear, bias table, qkv bias incl. the padded-token path)."""
    from nerf_rpn_amd import lib
    from nerf_rpn_amd.model.feature_extractor import ShiftedWindowAttention
    b, h, w, d = shape
    c = 32 * heads
    att = ShiftedWindowAttention(c, [4, 4, 4], [shift] * 3, heads)
    seeded_state(att, 3)
    att = att.to(dev)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(b, h, w, d, c, generator=g).to(dev).bfloat16()
    dy = torch.randn(b, h, w, d, c, generator=g).to(dev).bfloat16()
    res = {}
    for mode in (0, 1):
        lib.call("set_window_attn_mfma", mode)
        xg = x.clone().requires_grad_()
        y = att(xg)
        grads = torch.autograd.grad(y, [xg] + list(att.parameters()), dy)
        res[mode] = [y.float()] + [t.float() for t in grads]
    lib.call("set_window_attn_mfma", 1)
    names = ["y", "x"] + [n for n, _ in att.named_parameters()]
    for n, a, r in zip(names, res[1], res[0]):
        assert rel(a, r) < 3e-2, (n, rel(a, r))


def test_swin_rpn_two_runs_give_bit_identical_gradients(dev):
    """Swin-S + RPN (OBB) on a grid whose coarser stages need window padding: bit-identical gradients run to run (ordered per-unit
    partials for the relative-position table AND the padded-token bias gradient; no floating-point atomics across workgroups)."""
    from test_gpu_e2e import build, scene
    m = build(True, 160, dev, backbone="swin", sd=0.0).train()
    m.set_compute_dtype(torch.bfloat16)
    x = scene((40, 36, 44), 5).to(dev)
    gt = torch.tensor([[20., 18., 16., 14., 12., 10., 0.3], [12., 24., 30., 10., 9., 12., -0.8]], device=dev)
    labels = None
    grads = []
    for _ in range(2):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(1)                      # same anchor sample in both runs
        _, ls, _ = m([x], [gt])
        (ls["loss_objectness"] + 5.0 * ls["loss_rpn_box_reg"]).backward()
        grads.append(torch.cat([p.grad.reshape(-1).float() for p in m.parameters() if p.grad is not None]))
    assert torch.equal(grads[0], grads[1]), (grads[0] - grads[1]).abs().max().item()

"""GPU parity of rotated 3D RoIAlign (csrc/roialign.hip) with the oracle's C restatement of the reference CUDA op
(oracle/roialign.c == ROIAlignRotated3D_cuda.cu:13-343): forward to 1e-5 in fp32 (same operation order; device vs host sinf/cosf), backward within the
fixed-point resolution, deterministic across runs; the nn.Module keeps the reference's call contract."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_rois(R, N, dims, scale, seed):
    g = torch.Generator().manual_seed(seed)
    W, L, H = dims
    rois = torch.zeros(R, 8)
    rois[:, 0] = torch.randint(0, N, (R,), generator=g).float()
    ext = torch.tensor([W, L, H], dtype=torch.float32) / scale
    rois[:, 1:4] = torch.rand(R, 3, generator=g) * ext * 1.2 - 0.1 * ext           # some centres outside the grid
    rois[:, 4:7] = torch.rand(R, 3, generator=g) * ext * 0.8 + 0.3                 # incl. malformed (< 1 voxel) extents
    rois[:, 7] = (torch.rand(R, generator=g) - 0.5) * 180
    return rois


@pytest.mark.parametrize("cfg", [(2, 8, (9, 7, 6), (3, 3, 3), 0, 0.5), (1, 4, (6, 6, 6), (2, 3, 2), 2, 1.0), (1, 256, (20, 20, 17), (3, 3, 3), 0, 0.125),
                                 (2, 64, (10, 8, 5), (1, 1, 1), 0, 0.25)])
def test_forward_backward_match_oracle(cfg, dev):
    from nerf_rpn_amd.model.rotated_align import ROIAlignRotated3D
    from oracle import roialign as OR
    N, C, dims, pooled, samp, scale = cfg
    x = torch.randn(N, C, *dims, generator=torch.Generator().manual_seed(1))
    rois = make_rois(33, N, dims, scale, 2)
    ref = OR.roi_align_rotated_3d_forward(x, rois, scale, pooled, samp)
    xd = x.to(dev).requires_grad_(True)
    mod = ROIAlignRotated3D(list(pooled), samp)
    out = mod(xd, rois.to(dev), scale)
    assert tuple(out.shape) == (33, C, *pooled)
    # same fp32 operation order as the oracle; the only difference is the last ulp of the device's sinf / cosf in the sample positions
    assert torch.allclose(out.detach().cpu(), ref, atol=1e-5, rtol=1e-5), (out.detach().cpu() - ref).abs().max()
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    out.backward(g.to(dev))
    gref = OR.roi_align_rotated_3d_backward(g, rois, scale, pooled, (N, C, *dims), samp)
    err = (xd.grad.cpu().double() - gref).abs().max().item()
    assert err <= 1e-5 * max(1.0, gref.abs().max().item()), err
    # deterministic: integer (fixed-point) accumulation does not depend on the order of the atomics
    xd2 = x.to(dev).requires_grad_(True)
    mod(xd2, rois.to(dev), scale).backward(g.to(dev))
    assert torch.equal(xd.grad, xd2.grad)


def test_channels_last_bf16_and_known_answers(dev):
    from nerf_rpn_amd import ops
    from nerf_rpn_amd.model.rotated_align import roi_align_rotated_3d
    # theta = 0, a RoI covering the whole 4^3 grid, 2 samples per bin per axis: bins average the linear ramp at their sample centroid
    x = torch.arange(64, dtype=torch.float32).reshape(1, 1, 4, 4, 4).repeat(1, 4, 1, 1, 1)
    r = torch.tensor([[0, 2., 2., 2., 4., 4., 4., 0.]])
    o = roi_align_rotated_3d(x.to(dev), r.to(dev), [2, 2, 2], 1.0, 2).cpu()
    assert torch.allclose(o[0, 0].reshape(-1), torch.tensor([21., 22.75, 28., 29.75, 49., 50.75, 56., 57.75]))
    # a constant map stays constant under any rotation; rotating the RoI by 360 degrees changes nothing
    c = torch.full((1, 8, 7, 6, 5), 3.0, device=dev)
    rr = torch.tensor([[0, 3., 2.5, 2., 3., 2., 2., 33.]], device=dev)
    assert torch.allclose(roi_align_rotated_3d(c, rr, [3, 3, 3], 1.0, 0), torch.full((1, 8, 3, 3, 3), 3.0, device=dev))
    f = torch.randn(1, 8, 7, 6, 5, device=dev)
    rr2 = rr.clone(); rr2[0, 7] += 360.0
    assert torch.allclose(roi_align_rotated_3d(f, rr, [3, 3, 3], 1.0, 0), roi_align_rotated_3d(f, rr2, [3, 3, 3], 1.0, 0), atol=1e-4)
    # bf16 channels-last feature maps (what the bf16 backbone hands over) are consumed without a layout conversion
    fcl = torch.randn(1, 7, 6, 5, 16, device=dev).bfloat16()
    ob = roi_align_rotated_3d(fcl.permute(0, 4, 1, 2, 3), rr, [3, 3, 3], 1.0, 0)
    of = roi_align_rotated_3d(fcl.float().permute(0, 4, 1, 2, 3), rr, [3, 3, 3], 1.0, 0)
    assert ob.dtype == torch.bfloat16 and torch.allclose(ob.float(), of, atol=2e-2, rtol=2e-2)

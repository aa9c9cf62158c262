"""GPU parity of the fused balanced positive / negative sampler (csrc/postproc.hip nrpn_sample_pos_neg) against the CPU oracle
(oracle/sampler.py: the reference's counting / membership rules, model/utils.py:35-98, plus the kernel's k-smallest-key draw restated in
numpy -- bit-exact), its statistical uniformity (the reference draws with torch.randperm), and its use inside the training step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_labels(total, n_pos, n_ignore, seed):
    r = np.random.RandomState(seed)
    lab = np.zeros(total, dtype=np.float32)
    perm = r.permutation(total)
    lab[perm[:n_pos]] = 1.0
    lab[perm[n_pos:n_pos + n_ignore]] = -1.0
    return lab


CASES = [  # total, positives, ignored, batch, max_pos
    (950625, 300, 5000, 256, 128),      # the 160^3 scene's anchor count: more candidates than the batch in both classes
    (950625, 17, 100, 256, 128),        # few positives: negatives fill the batch (256 - 17)
    (950625, 0, 0, 256, 128),           # no positive at all
    (4000, 4000, 0, 256, 128),          # no negative at all
    (100, 10, 60, 256, 128),            # fewer candidates than the batch: everything is taken
    (1, 0, 0, 256, 128),
    (4200000, 5000, 100000, 512, 256),  # ~1000 candidates per key bin
    (70000, 30000, 0, 8192, 4096),      # the largest batch the kernel accepts
    (50000, 200, 49800, 256, 128),      # every non-positive ignored
]


@pytest.mark.parametrize("total,n_pos,n_ign,batch,max_pos", CASES)
def test_sampler_bit_exact_vs_oracle(total, n_pos, n_ign, batch, max_pos, dev):
    from nerf_rpn_amd import ops
    from oracle import sampler as OS
    lab = make_labels(total, n_pos, n_ign, total % 97)
    t = torch.from_numpy(lab).to(dev)
    for seed in (0, 1, 0x1234567890ABCDE, 2 ** 62 - 1):
        (pairs, extra) = ops.sample_pos_neg([t], batch, max_pos, seed, [torch.tensor(True, device=dev), torch.tensor(0, device=dev)])
        pos, neg = pairs[0]
        ep, en = OS.sample_pos_neg(lab, batch, max_pos, seed)
        assert pos.dtype == torch.int64 and neg.dtype == torch.int64
        assert np.array_equal(pos.cpu().numpy(), ep), (seed, pos.numel(), ep.size)
        assert np.array_equal(neg.cpu().numpy(), en), (seed, neg.numel(), en.size)
        assert extra == [1, 0]
        # the reference's rules (model/utils.py:68-77), independent of the draw
        assert pos.numel() == min(n_pos, max_pos) and neg.numel() == min(int((lab == 0).sum()), batch - pos.numel())
        assert (lab[pos.cpu().numpy()] >= 1).all() and (lab[neg.cpu().numpy()] == 0).all()


def test_sampler_batch_of_scenes_and_seed_dependence(dev):
    from nerf_rpn_amd import ops
    from oracle import sampler as OS
    labs = [make_labels(30000, 400, 300, s) for s in range(3)]
    ts = [torch.from_numpy(v).to(dev) for v in labs]
    a, _ = ops.sample_pos_neg(ts, 256, 128, 77)
    b, _ = ops.sample_pos_neg(ts, 256, 128, 77)
    c, _ = ops.sample_pos_neg(ts, 256, 128, 78)
    for i in range(3):
        assert torch.equal(a[i][0], b[i][0]) and torch.equal(a[i][1], b[i][1])          # a function of (labels, seed) only
        ep, en = OS.sample_pos_neg(labs[i], 256, 128, 77 + i)                             # scene i draws with seed + i
        assert np.array_equal(a[i][0].cpu().numpy(), ep) and np.array_equal(a[i][1].cpu().numpy(), en)
    assert not torch.equal(a[0][1], c[0][1])
    with pytest.raises(Exception):
        ops.sample_pos_neg(ts, 9000, 128, 1)                                              # batch beyond the kernel's capacity


def test_sampler_is_uniform(dev):
    """Every negative must be drawn with probability k / #neg (the reference draws with randperm): chi-square over 600 seeds."""
    from nerf_rpn_amd import ops
    total, n_pos = 3000, 40
    lab = make_labels(total, n_pos, 500, 5)
    t = torch.from_numpy(lab).to(dev)
    hits = np.zeros(total)
    draws = 600
    for seed in range(draws):
        (pairs, _) = ops.sample_pos_neg([t], 256, 128, seed * 7919 + 13)
        hits[pairs[0][1].cpu().numpy()] += 1
    neg = lab == 0
    k = 256 - n_pos
    p = k / neg.sum()
    assert hits[~neg].sum() == 0
    z = (hits[neg] - draws * p) / np.sqrt(draws * p * (1 - p))
    chi2 = float((z ** 2).sum())                   # ~ chi-square with #neg degrees of freedom (slightly under-dispersed: fixed k)
    dof = neg.sum()
    assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof), (chi2, dof)
    assert np.abs(z).max() < 5.5


def test_training_step_samples_once_and_keeps_the_box_check(dev, monkeypatch):
    """model.forward in training mode: the sampler runs through the fused kernel, the result is reproducible under torch.manual_seed,
    and a degenerate ground-truth box still raises the reference's assertion (nerf_rpn.py check_bbox_degeneration)."""
    import bench
    from nerf_rpn_amd import ops
    model = bench.build_model(torch.bfloat16, dev, "vgg")
    model.train()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4, 64, 64, 48, generator=g).to(dev)
    gt = torch.tensor([[20., 22., 18., 16., 12., 10., 0.3], [40., 30., 24., 12., 18., 14., -0.7]], device=dev)
    calls = []
    orig = ops.call

    def spy(name, *a):
        calls.append(name)
        return orig(name, *a)
    monkeypatch.setattr(ops, "call", spy)
    torch.manual_seed(11)
    model([x], [gt])
    first = {k: v.clone() for k, v in model.rpn.last_aux.items() if k in ("pos", "neg")}
    assert calls.count("sample_pos_neg") == 1
    torch.manual_seed(11)
    model([x], [gt])
    assert torch.equal(first["pos"], model.rpn.last_aux["pos"]) and torch.equal(first["neg"], model.rpn.last_aux["neg"])
    model([x], [gt])
    assert not torch.equal(first["neg"], model.rpn.last_aux["neg"])
    labels = model.rpn.last_aux["labels"][0]
    assert (labels[first["pos"]] >= 1).all() and (labels[first["neg"]] == 0).all()
    bad = gt.clone()
    bad[1, 4] = 0.0
    with pytest.raises(AssertionError, match="positive height, width and depth"):
        model([x], [bad])


def test_host_ground_truth_takes_the_prep_stream_and_gives_the_same_step(dev):
    """Ground truth handed over as host tensors: upload + matcher + sampler run on the model's own stream (no wait for the main stream's
    backlog); sampled anchors, losses and gradients must equal the device-resident path under the same seed, and the degenerate-box
    assertion still fires."""
    import bench
    model = bench.build_model(torch.bfloat16, dev, "vgg")
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4, 64, 48, 64, generator=g).to(dev)
    gt = torch.tensor([[20., 22., 18., 16., 12., 10., 0.3], [40., 30., 24., 12., 18., 14., -0.7], [30., 12., 40., 9., 9., 20., 1.1]])
    res = []
    for host in (False, True):
        torch.manual_seed(21)
        model.zero_grad(set_to_none=True)
        _, losses, _ = model([x], [gt if host else gt.to(dev)])
        (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]).backward()
        torch.cuda.synchronize()
        res.append((model.rpn.last_aux["pos"].clone(), model.rpn.last_aux["neg"].clone(), {k: v.detach().clone() for k, v in losses.items()},
                    torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None])))
    assert model._prep_stream is not None
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
    assert torch.equal(res[0][3], res[1][3])
    bad = gt.clone()
    bad[2, 3] = -1.0
    with pytest.raises(AssertionError, match="positive height, width and depth"):
        model([x], [bad])


def test_batch_of_two_scenes_samples_every_scene(dev):
    """Two scenes of different sizes (zero-padded to a common grid) through the training forward with the kernel sampler: per-scene
    counts follow the reference's rules, sampled anchors carry the right labels, padded anchors are never drawn, and the backward runs."""
    import bench
    model = bench.build_model(torch.bfloat16, dev, "vgg")
    g = torch.Generator().manual_seed(9)
    xs = [torch.rand(4, 64, 64, 48, generator=g).to(dev), torch.rand(4, 48, 56, 64, generator=g).to(dev)]
    gts = [torch.tensor([[20., 22., 18., 16., 12., 10., 0.3], [40., 30., 24., 12., 18., 14., -0.7]]),
           torch.tensor([[24., 20., 30., 14., 14., 12., 0.9]])]
    torch.manual_seed(4)
    _, losses, _ = model(xs, gts)
    loss = losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]
    loss.backward()
    assert torch.isfinite(loss)
    aux = model.rpn.last_aux
    labels = torch.cat(aux["labels"])
    T = aux["labels"][0].numel()
    pos, neg = aux["pos"], aux["neg"]
    assert (labels[pos] >= 1).all() and (labels[neg] == 0).all()
    assert (pos[1:] > pos[:-1]).all() and (neg[1:] > neg[:-1]).all()
    for i in range(2):
        lab = aux["labels"][i]
        n_pos = int(((pos >= i * T) & (pos < (i + 1) * T)).sum())
        n_neg = int(((neg >= i * T) & (neg < (i + 1) * T)).sum())
        assert n_pos == min(int((lab >= 1).sum()), 128) and n_neg == min(int((lab == 0).sum()), 256 - n_pos), (i, n_pos, n_neg)
    assert int((aux["labels"][1] < 0).sum()) > 0          # the smaller scene has ignored (padding) anchors

"""Does the dtype the headline is quoted in TRAIN like fp32?  (VERDICT r3 #5.)

tests/golden/convergence_obb_160.npz holds 40 optimiser steps of the REFERENCE's training loop (run_rpn.py:345-349, 373-395: AdamW 1e-4 / wd 0.01,
OneCycleLR over the run, clip_grad_norm 0.1, loss = objectness + 5 x box regression) on 4 fixed synthetic 160^3 scenes with the fixture
weights -- the bench workload -- run in the build container (make_golden.py::gen_convergence): the three losses of every step and the
anchors its sampler drew.  Here the HIP trainer repeats the run with the same draws injected (labels depend on anchors and ground truth
only), once in fp32 and once in bf16:
  * fp32 HIP against the reference: the first steps to 1e-4 (identical weights), the whole curve within a band (random init + train-mode
    BatchNorm at batch 1 amplify rounding differences step by step: two fp32 runs of the same algorithm drift apart as well),
  * bf16 HIP against fp32 HIP: every step inside the band, the mean of the last 8 steps within 5 %.
The bands are stated below with the measured values next to them."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dtype, g, dev, steps):
    from test_gpu_e2e import build, scene
    from nerf_rpn_amd.engine import FlatTrainer
    m = build(True, 160, dev).train()
    m.set_compute_dtype(dtype)
    trainer = FlatTrainer(m, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, total_steps=int(g["steps"]))
    nscene = int(g["scenes"])
    xs = [scene((160, 160, 160), 300 + i).to(dev) for i in range(nscene)]
    gts = [torch.from_numpy(g[f"gt{i}"]).to(dev) for i in range(nscene)]
    pos_all, neg_all = torch.from_numpy(g["pos"]).to(dev), torch.from_numpy(g["neg"]).to(dev)
    po, no = g["pos_off"], g["neg_off"]
    out = []
    for it in range(steps):
        k = it % nscene
        pos, neg = pos_all[po[it]:po[it + 1]], neg_all[no[it]:no[it + 1]]
        m.rpn.sampler_hook = lambda labels, pos=pos, neg=neg: (pos, neg)
        _, losses, _ = m([xs[k]], [gts[k]])
        (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"] + 0.0 * losses["loss_rpn_box_reg_2d"]).backward()
        trainer.step()
        out.append([losses["loss_objectness"].item(), losses["loss_rpn_box_reg"].item()])
    return np.asarray(out)


def test_bf16_trains_like_fp32_and_like_the_reference(golden, dev):
    g = golden("convergence_obb_160")
    steps = int(g["steps"])
    ref = g["losses"][:, :2]
    f32 = _run(torch.float32, g, dev, steps)
    b16 = _run(torch.bfloat16, g, dev, steps)
    total = lambda a: a[:, 0] + 5.0 * a[:, 1]
    tr, t32, t16 = total(ref), total(f32), total(b16)
    d32 = np.abs(t32 - tr) / tr
    d16 = np.abs(t16 - t32) / t32
    print("[convergence] step: reference / fp32 HIP / bf16 HIP total loss")
    for i in range(steps):
        print(f"   {i:2d}: {tr[i]:8.4f} {t32[i]:8.4f} {t16[i]:8.4f}   fp32 vs ref {100 * d32[i]:5.2f} %   bf16 vs fp32 {100 * d16[i]:5.2f} %")
    tail = slice(steps - 8, steps)
    print(f"[convergence] worst step: fp32 vs reference {100 * d32.max():.2f} %, bf16 vs fp32 {100 * d16.max():.2f} %; last-8 mean: ref {tr[tail].mean():.4f} "
          f"fp32 {t32[tail].mean():.4f} bf16 {t16[tail].mean():.4f}")
    # identical weights on the first step: the parity tolerance of the train fixtures
    assert abs(f32[0, 0] - ref[0, 0]) < 1e-4 * max(1.0, ref[0, 0]) and abs(f32[0, 1] - ref[0, 1]) < 1e-4 * max(1.0, ref[0, 1])
    # the loss has to come down as it does in the reference (9.8 -> 2.1 over the run)
    assert t32[tail].mean() < 0.35 * t32[0] and t16[tail].mean() < 0.35 * t16[0]
    # Bands.  Measured (round 4): fp32 HIP follows the reference to <= 0.3 % for the first 8 steps, then the two fp32 runs drift apart
    # chaotically (random init, batch 1, train-mode BatchNorm: <= 17 % on single steps, 5.1 % on the mean of the last 8); bf16 against fp32
    # HIP: <= 1.6 % over the first 8 steps, <= 10.6 % on single steps, 3.7 % on the last-8 mean -- and 1.2 % from the reference's: bf16
    # sits INSIDE the spread of the two fp32 runs.
    assert d32[:8].max() < 0.01 and d16[:8].max() < 0.03, (d32[:8].max(), d16[:8].max())
    assert d32.max() < 0.25 and d16.max() < 0.20, (d32.max(), d16.max())
    m_ref, m32, m16 = tr[tail].mean(), t32[tail].mean(), t16[tail].mean()
    assert abs(m16 - m32) < 0.05 * m32 and abs(m16 - m_ref) < 0.05 * m_ref, (m_ref, m32, m16)
    assert abs(m32 - m_ref) < 0.08 * m_ref, (m_ref, m32)

"""Is the deviation of the bf16 features from the fp32 ones a property of the NETWORK or of the HIP kernels?  CPU experiment on the oracle
nets (plain torch, no HIP), with the fixtures' weights (tests/fixture_init.seeded_state, salt 1): fp32 against an emulation of bf16
STORAGE -- input, weights and the output of every conv / linear / norm / pool / activation / attention / block rounded to bf16, fp32
accumulation -- and, as the amplification of the network, against a run whose activations carry 1e-6 relative noise.
    python tools/bf16_chaos_cpu.py            regenerates tests/golden/bf16_emulation.json (all configurations, a few minutes)
    python tools/bf16_chaos_cpu.py resnet 64 56 48 train     one configuration, printed only"""
import json
import os
import sys

import torch
from torch import nn

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from fixture_init import seeded_state  # noqa: E402
from oracle import nets as ON  # noqa: E402

SWIN_S = dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24))
ROUNDED = (nn.Conv3d, nn.BatchNorm3d, nn.MaxPool3d, nn.ReLU, nn.Linear, nn.LayerNorm, nn.GELU, ON.WindowAttention, ON.SwinBlock)


def build(kind, train):
    if kind == "swin":
        m = ON.SwinFPN(SWIN_S["embed_dim"], SWIN_S["depths"], SWIN_S["num_heads"], 0.0)
    else:
        m = ON.ResNetFPN() if kind == "resnet" else ON.VGGFPN("EF", 4, 160)
    seeded_state(m, 1)
    return m.train() if train else m.eval()


def run(kind, train, x, mode):
    m = build(kind, train)
    if mode == "bf16":
        with torch.no_grad():
            for p in m.parameters():
                if p.dim() > 1:
                    p.copy_(p.bfloat16().float())
        rnd = lambda mod, inp, out: out.bfloat16().float()
    elif mode == "eps":
        g = torch.Generator().manual_seed(5)
        rnd = lambda mod, inp, out: out * (1.0 + 1e-6 * torch.randn(out.shape, generator=g))
    if mode != "fp32":
        for mod in m.modules():
            if isinstance(mod, ROUNDED):
                mod.register_forward_hook(rnd)
    with torch.no_grad():
        return [o.float() for o in m(x.bfloat16().float() if mode == "bf16" else x)]


def measure(kind, shape, train):
    x = torch.rand(1, 4, *shape, generator=torch.Generator().manual_seed(200))
    ref = run(kind, train, x, "fp32")
    out = {}
    for mode in ("eps", "bf16"):
        got = run(kind, train, x, mode)
        out[mode] = [round(((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item(), 5) for a, b in zip(ref, got)]
    return out


if len(sys.argv) > 4:
    print(measure(sys.argv[1], [int(v) for v in sys.argv[2:5]], not (len(sys.argv) > 5 and sys.argv[5] == "eval")))
else:
    res = {"note": "rms(fp32 - variant) / rms(fp32) of the four FPN outputs of the oracle nets on the CPU with the fixtures' weights "
                   "(tools/bf16_chaos_cpu.py). 'bf16': input, weights and every conv / linear / norm / pool / activation / attention / block "
                   "output rounded to bf16, fp32 accumulation. 'eps': 1e-6 relative noise on the same outputs (the network's amplification)."}
    for kind, shape, train in (("vgg", (48, 40, 32), True),            # small: recomputed by tests/test_oracle_golden.py (pins script <-> fixture)
                               ("vgg", (160, 160, 160), True), ("resnet", (160, 120, 64), True), ("swin", (80, 56, 48), True),
                               ("vgg", (160, 160, 160), False), ("resnet", (200, 200, 130), False), ("resnet", (160, 120, 64), False),
                               ("swin", (160, 120, 64), False), ("swin", (200, 200, 130), False)):
        key = f"{kind}_{shape[0]}x{shape[1]}x{shape[2]}" + ("" if train else "_eval")
        res[key] = measure(kind, shape, train)
        print(key, res[key], flush=True)
    path = os.path.join(root, "tests", "golden", "bf16_emulation.json")
    if os.path.exists(path):      # keep the 'grad/<fixture>' entries of tools/bf16_train_grad_cpu.py
        res.update({k: v for k, v in json.load(open(path)).items() if k.startswith("grad/")})
    json.dump(res, open(path, "w"), indent=1)

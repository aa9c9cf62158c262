"""Is the 10-50 % deviation of the bf16 train-mode features from the fp32 ones (tools/diag_bf16_train.py) a property of the NETWORK or of
the HIP kernels?  CPU experiment on the oracle nets (plain torch, no HIP): the same random-init backbone in train mode, fp32 against an
emulation of bf16 storage -- weights rounded to bf16, the output of every conv / norm / pool / add rounded to bf16, fp32 accumulation --
and, as the noise floor, against a run whose activations are perturbed by one fp32 ulp-scale relative noise (1e-6).
    python tools/bf16_chaos_cpu.py [resnet|vgg|swin] [X Y Z] [eval]"""
import os
import sys

import torch
from torch import nn

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
from oracle import nets as ON  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "resnet"
shape = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else [64, 56, 48]
torch.manual_seed(1)


SWIN_S = dict(embed_dim=96, depths=(2, 2, 18, 2), num_heads=(3, 6, 12, 24))
mode_train = not (len(sys.argv) > 5 and sys.argv[5] == "eval")


def build():
    torch.manual_seed(1)
    if kind == "swin":
        m = ON.SwinFPN(SWIN_S["embed_dim"], SWIN_S["depths"], SWIN_S["num_heads"], 0.0)
    else:
        m = ON.ResNetFPN() if kind == "resnet" else ON.VGGFPN("EF", 4, 160)
    return m.train() if mode_train else m.eval()


x = torch.rand(1, 4, *shape, generator=torch.Generator().manual_seed(200))


def run(mode):
    m = build()
    hooks = []
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
            if isinstance(mod, (nn.Conv3d, nn.BatchNorm3d, nn.MaxPool3d, nn.ReLU, nn.Linear, nn.LayerNorm, nn.GELU, ON.WindowAttention, ON.SwinBlock)):
                hooks.append(mod.register_forward_hook(rnd))
    with torch.no_grad():
        out = m(x.bfloat16().float() if mode == "bf16" else x)
    return [o.float() for o in out]


ref = run("fp32")
for mode in ("eps", "bf16"):
    got = run(mode)
    print(mode, [f"{((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item():.4f}" for a, b in zip(ref, got)], "(rms diff / rms of each FPN output)")

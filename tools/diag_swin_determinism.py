import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import test_gpu_graph as TG
dev = torch.device("cuda:0")
def golden(name):
    return dict(np.load(os.path.join("/root/repo/tests/golden", name + ".npz"), allow_pickle=False))
a = TG._run(dev, golden, False, "swin", torch.float32, steps=4)
b = TG._run(dev, golden, False, "swin", torch.float32, steps=4)
print("eager vs eager losses equal:", a[0] == b[0])
for i, (x, y) in enumerate(zip(a[1], b[1])):
    d = (x - y).abs()
    print(" step", i, "arena equal", torch.equal(x, y), "max diff", d.max().item(), "at", int(d.argmax()))

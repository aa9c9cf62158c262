"""Developer check without a GPU: compiles the device code of csrc/geomloss.hip (+ geometry.cuh) as HOST C++ (the kernel body turned
into a loop) and compares loss / gradient with the CPU oracle's autograd and the reference goldens.  Not part of the product or the
test suite -- the GPU tests in tests/test_gpu_geomloss.py are the parity evidence; this only shortens the edit loop.
    python tools/host_check_geomloss.py"""
import os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
geo = open(os.path.join(ROOT, "nerf_rpn_amd/csrc/geometry.cuh")).read().replace('#include "common.h"', '').replace('#pragma once', '')
src = open(os.path.join(ROOT, "nerf_rpn_amd/csrc/geomloss.hip")).read().replace('#include "geometry.cuh"', '')
pre = """#include <cmath>
#include <cstdint>
#include <cstdio>
#define __device__
#define __forceinline__ inline
#define __global__
#define __restrict__
typedef void* nrpn_stream_t;
static inline int64_t cdiv64(int64_t a,int64_t b){return (a+b-1)/b;}
#define NRPN_REQUIRE(c, ...) do{ if(!(c)) return -1; }while(0)
#define NRPN_LAUNCH_CHECK(x)
#define NRPN_OK 0
"""
src = src.replace("""  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[7], q[7];""", """  for (int64_t i = 0; i < n; ++i) {
  float p[7], q[7];""")
src = src.replace("""  for (int k = 0; k < 7; ++k) grad[i * 7 + k] = l.g[k];
}
""", """  for (int k = 0; k < 7; ++k) grad[i * 7 + k] = l.g[k];
  }
}
""")
src = re.sub(r"hipLaunchKernelGGL\(rotated_iou_loss_kernel,[^;]*;", "rotated_iou_loss_kernel(pred, target, n, mode, loss, grad, iou);", src, flags=re.S)
open(os.path.join(tmp, "host.cpp"), "w").write(pre + geo + src)
subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-o", os.path.join(tmp, "host.so"), os.path.join(tmp, "host.cpp")])
import ctypes, torch, numpy as np, math, sys
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
lib=ctypes.CDLL(os.path.join(tmp, 'host.so'))
from oracle import geometry as OG
g=dict(np.load(os.path.join(ROOT, 'tests/golden/geometry.npz')))
def run(b1,b2,mode):
    n=b1.shape[0]
    p=b1.contiguous().numpy().astype(np.float32); t=b2.contiguous().numpy().astype(np.float32)
    loss=np.zeros(n,np.float32); grad=np.zeros((n,7),np.float32); iou=np.zeros(n,np.float32)
    rc=lib.nrpn_rotated_iou_loss_f32(p.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n), mode, loss.ctypes.data_as(ctypes.c_void_p), grad.ctypes.data_as(ctypes.c_void_p), iou.ctypes.data_as(ctypes.c_void_p), None)
    assert rc==0
    return torch.from_numpy(loss), torch.from_numpy(grad), torch.from_numpy(iou)
def oracle_loss(mode,b1,b2):
    if mode in (0,1):
        iou,_,_,_,u=OG.iou_3d(b1,b2,verbose=True); r=(iou*u+1)/(u+1)
        return -torch.log(r) if mode==0 else 1-r
    if mode==2: return OG.giou_3d(b1,b2)[0]
    return OG.diou_3d(b1,b2)[0]
def rand_pairs(n, seed):
    gg = torch.Generator().manual_seed(seed)
    c = torch.rand(n, 3, generator=gg) * 20 + 10
    s = torch.rand(n, 3, generator=gg) * 18 + 2
    t = (torch.rand(n, 1, generator=gg) - 0.5) * math.pi
    b1 = torch.cat([c, s, t], dim=1)
    c2 = c + (torch.rand(n, 3, generator=gg) - 0.5) * s * 1.2
    s2 = s * (0.5 + torch.rand(n, 3, generator=gg))
    t2 = t + (torch.rand(n, 1, generator=gg) - 0.5) * 1.5
    far = torch.rand(n, generator=gg) < 0.15
    c2[far] = c2[far] + 60
    return b1, torch.cat([c2, s2, t2], dim=1)
B1,B2=torch.from_numpy(g['b1'])[0],torch.from_numpy(g['b2'])[0]
for mode in range(4):
    for (b1,b2) in [(B1[9:209],B2[9:209]), rand_pairs(600,3)]:
        l,gr,iou=run(b1,b2,mode)
        c=b1.clone().requires_grad_(True)
        lo=oracle_loss(mode,c.unsqueeze(0),b2.unsqueeze(0))[0]
        lo.sum().backward()
        gerr=(gr-c.grad).abs()
        bad=(gerr>1e-4+1e-3*c.grad.abs()).any(dim=1)
        print(mode, 'loss err',(l-lo.detach()).abs().max().item(),'grad err max',gerr.max().item(),'bad pairs',int(bad.sum()), 'nan', int(torch.isnan(gr).sum()))
l,gr,iou=run(B1,B2,2); print('giou golden err',(l-torch.from_numpy(g['giou_loss'])[0]).abs().max().item(), (iou-torch.from_numpy(g['iou3d'])[0]).abs().max().item())
l,gr,iou=run(B1,B2,3); print('diou golden err',(l-torch.from_numpy(g['diou_loss'])[0]).abs().max().item())

"""Window-attention kernels on the four Swin-S stage geometries of a 160^3 scene: VALU vs MFMA (bf16), forward and backward."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from nerf_rpn_amd import lib, ops
dev = torch.device('cuda:0')
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
t = torch.arange(64); code = (t // 16) * 49 + ((t // 4) % 4) * 7 + t % 4
idx = (code[:, None] - code[None, :] + 171).to(torch.int32).to(dev).contiguous()
for grid, heads in [(40, 3), (20, 6), (10, 12), (5, 24)]:
    c = 32 * heads
    qkv = torch.randn(1, grid, grid, grid, 3 * c, device=dev).bfloat16()
    dout = torch.randn(1, grid, grid, grid, c, device=dev).bfloat16()
    qb = torch.randn(3 * c, device=dev); tab = torch.randn(343, heads, device=dev) * 0.1
    out = torch.empty(1, grid, grid, grid, c, device=dev, dtype=torch.bfloat16)
    dqkv = torch.empty_like(qkv); dtab = torch.empty_like(tab); dpad = torch.empty(3 * c, device=dev)
    ws = torch.empty(lib.query('window_attn_bwd_workspace_bytes', 1, grid, grid, grid, heads), dtype=torch.uint8, device=dev)
    line = f'{grid}^3 heads {heads}:'
    for shift in (0, 1):
        for mfma in (0, 1):
            lib.call('set_window_attn_mfma', mfma)
            tf = timeit(lambda: lib.call('window_attn_fwd', qkv.data_ptr(), qb.data_ptr(), tab.data_ptr(), idx.data_ptr(), out.data_ptr(), 1, grid, grid, grid, c, heads, shift, 1, ops._s()))
            tb = timeit(lambda: lib.call('window_attn_bwd', qkv.data_ptr(), qb.data_ptr(), tab.data_ptr(), idx.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dtab.data_ptr(), dpad.data_ptr(), 1, grid, grid, grid, c, heads, shift, 1, ws.data_ptr(), ops._s()))
            line += f'  s{shift} {"mfma" if mfma else "valu"}: fwd {tf:.0f} bwd {tb:.0f} us'
    print(line)
lib.call('set_window_attn_mfma', 1)

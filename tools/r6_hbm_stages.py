"""The HBM-bound companions on their largest shapes (bench.py hbm_stages), stand-alone:  python tools/r6_hbm_stages.py [pool_fast=1]"""
import json, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from nerf_rpn_amd import lib
if len(sys.argv) > 1:
    lib.call("set_pool_fast", int(sys.argv[1]))
for e in bench.hbm_stages(torch.bfloat16, torch.device("cuda:0")):
    print(f"{e['kernel']:20s} {e['shape']:34s} {e['avg_us']:7.1f} us  {e['gbs']:7.1f} GB/s  {e['frac']:.3f}")

"""ORACLE (test infrastructure only): rotated 3D RoIAlign through ``roialign.c`` (plain C restatement of reference
nerf_rpn/model/rotated_align/src/cuda_3d/ROIAlignRotated3D_cuda.cu:13-343).  PARITY UNPINNED -- see the header of roialign.c.
Layouts are the reference op's: input [N,C,W,L,H] f32, rois [R,8], output [R,C,pw,pl,ph]."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_roialign.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
    return _LIB


def _args(inp_shape, rois, spatial_scale, output_size, sampling_ratio):
    n, c, w, l, h = inp_shape
    return [ctypes.c_int64(rois.shape[0]), ctypes.c_int(c), ctypes.c_int(w), ctypes.c_int(l), ctypes.c_int(h), ctypes.c_float(spatial_scale),
            ctypes.c_int(output_size[0]), ctypes.c_int(output_size[1]), ctypes.c_int(output_size[2]), ctypes.c_int(sampling_ratio)]


def roi_align_rotated_3d_forward(inp, rois, spatial_scale, output_size, sampling_ratio):
    x = np.ascontiguousarray(inp.detach().numpy().astype(np.float32))
    r = np.ascontiguousarray(rois.detach().numpy().astype(np.float32))
    out = np.zeros((r.shape[0], x.shape[1], *output_size), np.float32)
    _lib().oracle_roi_align_rotated_3d_fwd(x.ctypes.data_as(ctypes.c_void_p), r.ctypes.data_as(ctypes.c_void_p),
                                           *_args(x.shape, r, spatial_scale, output_size, sampling_ratio), out.ctypes.data_as(ctypes.c_void_p))
    return torch.from_numpy(out)


def roi_align_rotated_3d_backward(grad, rois, spatial_scale, output_size, inp_shape, sampling_ratio):
    """-> float64 [N,C,W,L,H] (the serial sum of the reference's float contributions, accumulated in double)."""
    g = np.ascontiguousarray(grad.detach().numpy().astype(np.float32))
    r = np.ascontiguousarray(rois.detach().numpy().astype(np.float32))
    out = np.zeros(tuple(inp_shape), np.float64)
    _lib().oracle_roi_align_rotated_3d_bwd(g.ctypes.data_as(ctypes.c_void_p), r.ctypes.data_as(ctypes.c_void_p),
                                           *_args(inp_shape, r, spatial_scale, output_size, sampling_ratio), out.ctypes.data_as(ctypes.c_void_p))
    return torch.from_numpy(out)

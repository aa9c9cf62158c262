"""nerf_rpn_amd -- MI355X-native (gfx950) engine for the 3D RPN-over-NeRF hot path of lyclyc52/NeRF_RPN.

Python host code mirrors the reference's model API (``nerf_rpn_amd.model.*``) and calls hand-written HIP kernels
through the C ABI in ``include/nerfrpn.h`` (``libnerfrpn_hip.so``).  No CPU fallback exists on the product path.
"""
from . import lib  # noqa: F401

__all__ = ["lib"]

#!/usr/bin/env python
"""Instruction mix of the MFMA loops of the conv kernels, from the compiler's ISA (no GPU needed).

    python tools/isa_audit.py                       # the kernels a training step spends its time in
    python tools/isa_audit.py conv3d.hip stem_wgrad  # any kernel of a source file whose mangled name contains the given substrings

For every kernel: VGPRs / AGPRs / scratch, and for each innermost loop that issues MFMAs the count of MFMA, other VALU, SALU, LDS, vector
memory, s_waitcnt and barrier instructions.  MFMA, VALU and the LDS / VMEM address arithmetic share a SIMD's issue slots: instructions
that are not MFMAs are not free in an "MFMA-bound" kernel (the integer bf16 rounding of the epilogues cost 1.5 % of the
dominant kernel, three 64-bit divisions per chunk made the stem's weight-gradient kernel issue-bound)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf_rpn_amd", "csrc")
DEFAULT = [("conv_halo.hip", ["conv_halo_kernelILi0"]),
           ("conv3d.hip", ["conv_wgrad_big_kernelILb0", "conv_igemm_big_kernelILb0ELb1ELi0ELb0", "conv_igemm_kernelItLi128ELi0ELb0ELi128ELb1ELi128ELb0",
                           "conv_igemm_kernelItLi128ELi0ELb0ELi128ELb1ELi128ELb1", "conv_igemm_kernelItLi64ELi0ELb0ELi128ELb1ELi128ELb0",
                           "conv_wgrad_kernelItLi0ELb1ELb0", "stem_wgrad_zrow_kernelItE"])]


def isa(src):
    out = os.path.join(tempfile.gettempdir(), "isa_audit_" + os.path.basename(src) + ".s")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".cuh", ".h"))):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include"),
                               "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    return open(out).read()


def audit(text, want):
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)s_endpgm", text, re.S | re.M):
        name = m.group(1)
        if not any(w in name for w in want):
            continue
        lines = m.group(2).split("\n")
        label = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
        loops = []
        for i, l in enumerate(lines):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if mm and mm.group(1) in label and label[mm.group(1)] < i:
                loops.append((label[mm.group(1)], i))

        def ins(a, b):
            return [x.strip() for x in lines[a:b + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
        with_mfma = [(a, b) for a, b in loops if any("v_mfma" in x for x in ins(a, b))]
        inner = [(a, b) for a, b in with_mfma if not any(a2 >= a and b2 <= b and (a2, b2) != (a, b) for a2, b2 in with_mfma)]
        res = {k: (re.search(re.escape(name) + r"\." + k + r", (\d+)", text) or [None, "?"])[1] for k in ("num_vgpr", "num_agpr", "private_seg_size")}
        print(f"{name[:96]}  vgpr {res['num_vgpr']} agpr {res['num_agpr']} scratch {res['private_seg_size']}")
        for a, b in inner:
            I = ins(a, b)
            n = lambda f: sum(1 for x in I if f(x))
            mf = n(lambda x: "v_mfma" in x)
            va = n(lambda x: x.startswith("v_") and "v_mfma" not in x)
            sa = n(lambda x: x.startswith("s_") and not x.startswith(("s_waitcnt", "s_barrier", "s_nop")))
            print(f"    MFMA loop: {len(I)} instructions = {mf} MFMA + {va} VALU ({va / mf:.2f} per MFMA) + {sa} SALU + {n(lambda x: x.startswith('ds_'))} LDS + "
                  f"{n(lambda x: x.startswith(('buffer_', 'global_')))} VMEM + {n(lambda x: x.startswith('s_waitcnt'))} waitcnt + {n(lambda x: x.startswith('s_barrier'))} barrier")


if __name__ == "__main__":
    jobs = DEFAULT if len(sys.argv) < 3 else [(sys.argv[1], sys.argv[2:])]
    for src, want in jobs:
        audit(isa(src), want)

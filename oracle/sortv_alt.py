"""ORACLE (test infrastructure only): a SECOND, independently written statement of what ``sort_vertices`` computes, used to cross-check
``sortv.c`` -- neither of the two can be compared with the CUDA source itself (sort_vert_kernel.cu needs ATen + CUDA; the reference holds
no vectors for it), so the index-level behaviour stays "parity unpinned"; this narrows what can be wrong to what BOTH derivations share.

Derivation (from the kernel's documented intent, sort_vert_kernel.cu:11-14, not from its code path): the valid vertices, given relative
to their centroid, are listed counter-clockwise by polar angle in [0, 2 pi) -- the comparator puts y > 0 before y < 0, larger
|x| x / r^2 (= cos|cos|) first in the upper half plane and last in the lower one -- duplicates (the same point reached twice: a corner of
one box that is also an edge intersection, or the corners of two identical boxes) appear once, the list is closed with its first index and
padded with an invalid intersection slot.  Here: float64 ``atan2`` angles, a stable sort, then duplicate removal by coordinates.

``classify`` names the inputs on which the CUDA source is undefined or on which float32 comparisons are too close to call; the
cross-check test asserts equality on everything else and that every excluded input falls in one of the listed classes."""
import math

import numpy as np

EPS = 1e-8


def sort_rows(vertices, mask, num_valid):
    """vertices [K, M, 2] float, mask [K, M] bool, num_valid [K] int -> int32 [K, 9] (same contract as sort_vertices_forward)."""
    v = np.asarray(vertices, dtype=np.float64)
    m = np.asarray(mask).astype(bool)
    K, M = m.shape
    out = np.zeros((K, 9), dtype=np.int32)
    for i in range(K):
        inval = np.where(~m[i, 8:])[0]
        pad = 8 + int(inval[0]) if inval.size else M - 1
        nv = int(num_valid[i])
        if nv < 3:
            out[i, :] = pad
            continue
        nv = min(nv, 8)
        idx = np.where(m[i])[0]
        ang = np.mod(np.arctan2(v[i, idx, 1], v[i, idx, 0]), 2 * math.pi)
        order = idx[np.argsort(ang, kind="stable")]
        picked = []
        for k in order:                      # a point already listed (within the kernel's 1e-8 equality) is not listed again
            if any(abs(v[i, k, 0] - v[i, p, 0]) < EPS and abs(v[i, k, 1] - v[i, p, 1]) < EPS for p in picked):
                continue
            picked.append(int(k))
        row = picked[:nv]
        n = len(row)
        out[i, :n] = row
        if n < nv:                           # fewer distinct points than num_valid: the kernel keeps selecting index 0 (see classify)
            out[i, n:nv] = 0
        out[i, nv] = out[i, 0]
        out[i, nv + 1:] = pad
        if nv == 8:                          # identical boxes: each corner of box 2 coincides with one of box 1
            dup = sum(1 for a in out[i, :4] for b in out[i, 4:8] if a == b)
            if dup == 4:
                out[i, 4] = out[i, 0]
                out[i, 5:] = pad
    return out


def classify(vertices, mask, num_valid, ang_tol=1e-4):
    """Per row: '' = well defined for a float32 implementation, otherwise the reason it is excluded from the equality check."""
    v = np.asarray(vertices, dtype=np.float64)
    m = np.asarray(mask).astype(bool)
    out = []
    for i in range(m.shape[0]):
        nv = int(num_valid[i])
        idx = np.where(m[i])[0]
        why = ""
        if nv != idx.size:
            why = "num_valid != popcount(mask)"                         # the caller's contract (box_intersection_2d.py:141-143)
        elif nv > 8:
            why = "num_valid > 8 overflows the 9-slot row (B6)"
        elif m[i, 8:].all():
            why = "all 16 intersection slots valid: `pad` is read uninitialised (B6)"
        elif nv >= 3:
            x, y = v[i, idx, 0], v[i, idx, 1]
            if (np.abs(y) < 1e-6).any():
                why = "a vertex on the x axis: the comparator falls off its end (B6)"
            else:
                ang = np.sort(np.mod(np.arctan2(y, x), 2 * math.pi))
                gaps = np.diff(np.concatenate([ang, ang[:1] + 2 * math.pi]))
                if (gaps < ang_tol).any():
                    why = "two vertices closer than 1e-4 rad: float32 ordering / the 1e-8 equality are too close to call"
        out.append(why)
    return out

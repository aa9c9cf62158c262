// Device geometry for the RPN post-processing kernels: fused rotated 3D IoU (one lane per box pair, everything in
// registers), AABB IoU, anchor-from-index, and the two box coders.
//
// Behavioural spec = reference nerf_rpn/model/rotated_iou/{oriented_iou_loss.py:6-107, box_intersection_2d.py:11-176},
// cuda_op/sort_vert_kernel.cu:15-134, model/utils.py:418-458, model/anchor.py:49-122, coder/AABB_coder.py:7-137,
// coder/midpoint_offset_coder.py:106-222, coder/misc.py:3-93.  This translation unit is compiled with
// -ffp-contract=off so that the fp32 operation order written here is the one executed (comparisons against
// thresholds / epsilons must not see fused multiply-adds).
#pragma once
#include "common.h"

namespace geo {

constexpr float kPiRef = 3.141592f;           // misc.py:3 (truncated on purpose)
constexpr float kHalfPiRef = (float)(3.141592 / 2);

// ---- polygon vertex order (sort_vert_kernel.cu:15-40) -------------------------------------------------------
__device__ __forceinline__ bool vert_before(float x1, float y1, float x2, float y2) {
  // a float f satisfies (double)f < 1e-8 exactly when f <= (float)1e-8 (the float just below 1e-8), and
  // (double)f > 1e-8 exactly when f > (float)1e-8 -- so the reference's double-literal comparisons stay in fp32.
  const float e = (float)1e-8;
  if (fabsf(x1 - x2) <= e && fabsf(y2 - y1) <= e) return false;
  if (y1 > 0.f && y2 < 0.f) return true;
  if (y1 < 0.f && y2 > 0.f) return false;
  float n1 = (float)((double)(x1 * x1 + y1 * y1) + 1e-8);
  float n2 = (float)((double)(x2 * x2 + y2 * y2) + 1e-8);
  float d = fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2;
  if (y1 > 0.f && y2 > 0.f) return d > e;
  if (y1 < 0.f && y2 < 0.f) return d <= e;
  return false;  // y == 0 on either side: undefined in the reference, defined false
}

// The same predicate with the per-vertex term |x| x / (x^2 + y^2 + 1e-8) computed once per vertex (q1, q2) instead of inside every
// comparison of the selection sort (8 picks x 24 candidates x 2 comparisons, each with two divisions and an fp64 round trip): the
// operations and their order per term are those of vert_before, so d = q1 - q2 is the same float.
__device__ __forceinline__ float vert_q(float x, float y) {
  const float n = (float)((double)(x * x + y * y) + 1e-8);
  return fabsf(x) * x / n;
}
// Branch-free (the early returns of vert_before are mutually exclusive cases): one lane per box pair runs this 8 x 24 x 2 times, and
// as control flow it compiled to ~125 instructions and a dozen exec-mask branches per candidate.
__device__ __forceinline__ bool vert_before_q(float x1, float y1, float q1, float x2, float y2, float q2) {
  const float e = (float)1e-8;
  const bool same = (fabsf(x1 - x2) <= e) & (fabsf(y2 - y1) <= e);
  const bool up1 = y1 > 0.f, dn1 = y1 < 0.f, up2 = y2 > 0.f, dn2 = y2 < 0.f;
  const float d = q1 - q2;
  const bool r = (up1 & dn2) | (up1 & up2 & (d > e)) | (dn1 & dn2 & (d <= e));
  return (!same) & r;
}

__device__ __forceinline__ void corners2d(float cx, float cy, float w, float h, float a, float *X, float *Y) {
  const float s = sinf(a), c = cosf(a);
  const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f};
  const float sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float lx = sx[i] * w, ly = sy[i] * h;
    X[i] = (lx * c + ly * (-s)) + cx;
    Y[i] = (lx * s + ly * c) + cy;
  }
}

// corners P (4) inside box Q (box_intersection_2d.py:54-79)
__device__ __forceinline__ void inside4(const float *PX, const float *PY, const float *QX, const float *QY, bool *in) {
  const float abx = QX[1] - QX[0], aby = QY[1] - QY[0];
  const float adx = QX[3] - QX[0], ady = QY[3] - QY[0];
  const float nab = abx * abx + aby * aby, nad = adx * adx + ady * ady;
  const float hi = (float)(1 + 1e-6), lo = -1e-6f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float amx = PX[i] - QX[0], amy = PY[i] - QY[0];
    float r1 = (abx * amx + aby * amy) / nab;
    float r2 = (adx * amx + ady * amy) / nad;
    in[i] = (r1 > lo) && (r1 < hi) && (r2 > lo) && (r2 < hi);
  }
}

// Intersection area of two rotated rectangles given their corners (oriented_box_intersection_2d).
__device__ __forceinline__ float rect_intersection_area(const float *AX, const float *AY, const float *BX, const float *BY) {
  float ox[24], oy[24];
  bool ok[24];
#pragma unroll
  for (int i = 0; i < 4; ++i) { ox[i] = AX[i]; oy[i] = AY[i]; ox[4 + i] = BX[i]; oy[4 + i] = BY[i]; }
  inside4(AX, AY, BX, BY, ok);
  inside4(BX, BY, AX, AY, ok + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x1 = AX[i], y1 = AY[i], x2 = AX[(i + 1) & 3], y2 = AY[(i + 1) & 3];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x3 = BX[j], y3 = BY[j], x4 = BX[(j + 1) & 3], y4 = BY[(j + 1) & 3];
      const float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
      const float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
      const float den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3);
      bool v = false;
      if (num != 0.0f) {
        const float t = den_t / num, u = -den_u / num;
        v = (t > 0.f) && (t < 1.f) && (u > 0.f) && (u < 1.f);
      }
      const float ts = den_t / (num + 1e-8f);
      const int k = 8 + i * 4 + j;
      ok[k] = v;
      ox[k] = v ? x1 + ts * (x2 - x1) : 0.f;
      oy[k] = v ? y1 + ts * (y2 - y1) : 0.f;
    }
  }
  int nv = 0;
  float mx = 0.f, my = 0.f;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    nv += ok[k] ? 1 : 0;
    mx += ok[k] ? ox[k] : 0.f;
    my += ok[k] ? oy[k] : 0.f;
  }
  if (nv < 3) return 0.f;  // every slot is the zero pad vertex
  mx /= (float)nv;
  my /= (float)nv;
  float nx[24], ny[24], nq[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) { nx[k] = ox[k] - mx; ny[k] = oy[k] - my; nq[k] = vert_q(nx[k], ny[k]); }
  const float q_start = vert_q(1.0f, (float)(-1e-8));
  if (nv > 8) nv = 8;
  // Selection sort by angle (runtime loop over the <= 8 picks, the 24-candidate scan unrolled), with the shoelace sum
  // accumulated on the fly from the *original* coordinates of consecutive picks.
  float fx = 0.f, fy = 0.f;      // first pick (original coords)
  float qx = 0.f, qy = 0.f;      // previous pick (original coords)
  float px = 0.f, py = 0.f, pq = 0.f;      // previous pick (normalised coords, its vert_q)
  float total = 0.f, total4 = 0.f;
  int k0 = -1, k1 = -1, k2 = -1, k3 = -1, dup = 0;
#pragma unroll 1
  for (int j = 0; j < nv; ++j) {
    // (bx, by, bq) = best so far, starting from the reference's sentinel; until a candidate wins the pick is vertex 0
    float bx = 1.0f, by = (float)(-1e-8), bq = q_start;
    float tx = ox[0], ty = oy[0], tnx = nx[0], tny = ny[0], tq = nq[0];
    int take = 0;
    const bool first = (j == 0);
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const bool c = ok[k] & vert_before_q(nx[k], ny[k], nq[k], bx, by, bq) & (first | vert_before_q(px, py, pq, nx[k], ny[k], nq[k]));
      bx = c ? nx[k] : bx; by = c ? ny[k] : by; bq = c ? nq[k] : bq;
      tx = c ? ox[k] : tx; ty = c ? oy[k] : ty; take = c ? k : take;
    }
    if (take != 0) { tnx = bx; tny = by; tq = bq; }      // take == 0: nobody won, or vertex 0 won -- (tnx, tny) is vertex 0 either way
    if (j == 0) { fx = tx; fy = ty; k0 = take; }
    else total += qx * ty - qy * tx;
    if (j == 1) k1 = take;
    if (j == 2) k2 = take;
    if (j == 3) { k3 = take; total4 = total + (tx * fy - ty * fx); }
    if (j >= 4) dup += (take == k0) + (take == k1) + (take == k2) + (take == k3);
    qx = tx; qy = ty; px = tnx; py = tny; pq = tq;
  }
  total += qx * fy - qy * fx;  // close the polygon
  // identical boxes: the 8 corners coincide pairwise and the first four picks already are the polygon
  // (sort_vert_kernel.cu:110-129)
  if (nv == 8 && dup == 4) total = total4;
  return fabsf(total) / 2.f;
}

// 3D IoU of two z-rotated boxes (x,y,z,w,h,d,theta)  -- cal_iou_3d
__device__ __forceinline__ float iou3d_obb(const float *p, const float *q) {
  const float zt1 = p[2] + p[5] * 0.5f, zb1 = p[2] - p[5] * 0.5f;
  const float zt2 = q[2] + q[5] * 0.5f, zb2 = q[2] - q[5] * 0.5f;
  const float zov = fmaxf(fminf(zt1, zt2) - fmaxf(zb1, zb2), 0.f);
  float AX[4], AY[4], BX[4], BY[4];
  corners2d(p[0], p[1], p[3], p[4], p[6], AX, AY);
  corners2d(q[0], q[1], q[3], q[4], q[6], BX, BY);
  const float inter2 = rect_intersection_area(AX, AY, BX, BY);
  const float u2 = p[3] * p[4] + q[3] * q[4] - inter2;
  const float iou2 = inter2 / u2;
  const float inter3 = iou2 * u2 * zov;
  const float u3 = p[3] * p[4] * p[5] + q[3] * q[4] * q[5] - inter3;
  return inter3 / u3;
}

// IoU of two axis-aligned boxes (x1,y1,z1,x2,y2,z2) -- _aabb_inter_union_3d
__device__ __forceinline__ float iou3d_aabb(const float *a, const float *b) {
  const float va = (a[3] - a[0]) * (a[4] - a[1]) * (a[5] - a[2]);
  const float vb = (b[3] - b[0]) * (b[4] - b[1]) * (b[5] - b[2]);
  const float ex = fmaxf(fminf(a[3], b[3]) - fmaxf(a[0], b[0]), 0.f);
  const float ey = fmaxf(fminf(a[4], b[4]) - fmaxf(a[1], b[1]), 0.f);
  const float ez = fmaxf(fminf(a[5], b[5]) - fmaxf(a[2], b[2]), 0.f);
  const float inter = ex * ey * ez;
  return inter / (va + vb - inter);
}

template <int W> __device__ __forceinline__ float iou3d(const float *a, const float *b) {
  if (W == 6) return iou3d_aabb(a, b);
  return iou3d_obb(a, b);
}

// ---- anchors from the pyramid table (layout documented in nerfrpn.h) ----------------------------------------
struct AnchorCell {
  float box[6];
  int level, ix, iy, iz, a;
};

__device__ __forceinline__ AnchorCell anchor_at(const int32_t *tab, int64_t flat) {
  const int L = tab[0], A = tab[1];
  int l = 0;
  int64_t first = 0;
  for (int i = 0; i < L; ++i) {
    const int32_t *t = tab + 2 + 8 * i;
    int64_t f = ((int64_t)(uint32_t)t[6]) | ((int64_t)t[7] << 32);
    if (flat >= f) { l = i; first = f; }
  }
  const int32_t *t = tab + 2 + 8 * l;
  const int64_t local = flat - first;
  AnchorCell c;
  c.level = l;
  c.a = (int)(local % A);
  int64_t cell = local / A;
  c.iz = (int)(cell % t[2]);
  cell /= t[2];
  c.iy = (int)(cell % t[1]);
  c.ix = (int)(cell / t[1]);
  const float *base = reinterpret_cast<const float *>(tab + 2 + 8 * L) + ((int64_t)l * A + c.a) * 6;
  const float shx = (float)c.ix * (float)t[3], shy = (float)c.iy * (float)t[4], shz = (float)c.iz * (float)t[5];
  c.box[0] = shx + base[0]; c.box[1] = shy + base[1]; c.box[2] = shz + base[2];
  c.box[3] = shx + base[3]; c.box[4] = shy + base[4]; c.box[5] = shz + base[5];
  return c;
}

// ---- coders ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ void decode_aabb(const float *d, const float *an, float *out) {
  const float clip = 7.6009024595420822f;  // log(2000)
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float w = an[ax + 3] - an[ax];
    const float c = an[ax] + 0.5f * w;
    const float pc = d[ax] * w + c;
    const float pw = expf(fminf(d[ax + 3], clip)) * w;
    const float half = 0.5f * pw;
    out[ax] = pc - half;
    out[ax + 3] = pc + half;
  }
}

__device__ __forceinline__ void encode_aabb(const float *gt, const float *an, float *out) {
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float aw = an[ax + 3] - an[ax];
    const float ac = an[ax] + 0.5f * aw;
    const float gw = gt[ax + 3] - gt[ax];
    const float gc = gt[ax] + 0.5f * gw;
    out[ax] = (gc - ac) / aw;
    out[ax + 3] = logf(gw / aw);
  }
}

__device__ __forceinline__ float py_mod(float a, float b) {
  float r = fmodf(a, b);
  if (r != 0.f && ((r < 0.f) != (b < 0.f))) r += b;
  return r;
}

// delta_sp2bbox: 8 deltas + AABB anchor -> (x,y,z,w,h,d,theta)
__device__ __forceinline__ void decode_midpoint(const float *dl, const float *an, float *out) {
  const float lim = 4.1351665567423561f;  // |log(16/1000)|
  float g[3], s[3];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) {
    const float pc = (an[ax] + an[ax + 3]) * 0.5f;
    const float pw = an[ax + 3] - an[ax];
    const float dd = fminf(fmaxf(dl[ax + 3], -lim), lim);
    s[ax] = pw * expf(dd);
    g[ax] = pc + pw * dl[ax];
  }
  const float da = fminf(fmaxf(dl[6], -0.5f), 0.5f), db = fminf(fmaxf(dl[7], -0.5f), 0.5f);
  const float gx = g[0], gy = g[1], gw = s[0], gh = s[1];
  const float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
  const float ga = gx + da * gw, ga_ = gx - da * gw, gb = gy + db * gh, gb_ = gy - db * gh;
  float px[4] = {ga, x2, ga_, x1}, py[4] = {y1, gb, y2, gb_};
  float cxv[4], cyv[4], diag[4], mdiag = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cxv[i] = px[i] - gx; cyv[i] = py[i] - gy;
    diag[i] = sqrtf(cxv[i] * cxv[i] + cyv[i] * cyv[i]);
    mdiag = (i == 0) ? diag[0] : fmaxf(mdiag, diag[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float sc = mdiag / diag[i];
    px[i] = cxv[i] * sc + gx;
    py[i] = cyv[i] * sc + gy;
  }
  // rectpoly2obb + regular_obb (misc.py:5-47)
  const float th = atan2f(-(py[1] - py[0]), px[1] - px[0] + 1e-7f);
  const float cs = cosf(th), sn = sinf(th);
  const float mxv = (((px[0] + px[1]) + px[2]) + px[3]) / 4.f;
  const float myv = (((py[0] + py[1]) + py[2]) + py[3]) / 4.f;
  float rxmin = 0.f, rxmax = 0.f, rymin = 0.f, rymax = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float qx = px[i] - mxv, qy = py[i] - myv;
    const float rx = qx * cs + qy * (-sn), ry = qx * sn + qy * cs;
    if (i == 0) { rxmin = rxmax = rx; rymin = rymax = ry; }
    else { rxmin = fminf(rxmin, rx); rxmax = fmaxf(rxmax, rx); rymin = fminf(rymin, ry); rymax = fmaxf(rymax, ry); }
  }
  const float w = rxmax - rxmin, h = rymax - rymin;
  const bool wide = w > h;
  float tr = wide ? th : th + kHalfPiRef;
  tr = py_mod(tr + kHalfPiRef, kPiRef) - kHalfPiRef;
  out[0] = mxv; out[1] = myv; out[2] = g[2];
  out[3] = wide ? w : h; out[4] = wide ? h : w; out[5] = s[2];
  out[6] = tr;
}

// bbox2delta_sp: gt OBB [7] + AABB anchor -> 8 deltas
__device__ __forceinline__ void encode_midpoint(const float *gt, const float *an, float *out) {
  const float cx = gt[0], cy = gt[1], w = gt[3], h = gt[4], t = gt[6];
  const float cs = cosf(t), sn = sinf(t);
  const float bx = fabsf(w / 2 * cs) + fabsf(h / 2 * sn), by = fabsf(w / 2 * sn) + fabsf(h / 2 * cs);
  const float hx1 = cx - bx, hy1 = cy - by, hx2 = cx + bx, hy2 = cy + by;
  const float gx = (hx1 + hx2) * 0.5f, gy = (hy1 + hy2) * 0.5f, gw = hx2 - hx1, gh = hy2 - hy1;
  const float v1x = w / 2 * cs, v1y = -w / 2 * sn, v2x = -h / 2 * sn, v2y = -h / 2 * cs;
  const float xs[4] = {cx + v1x + v2x, cx + v1x - v2x, cx - v1x - v2x, cx - v1x + v2x};
  const float ys[4] = {cy + v1y + v2y, cy + v1y - v2y, cy - v1y - v2y, cy - v1y + v2y};
  float ymin = ys[0], xmax = xs[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) { ymin = fminf(ymin, ys[i]); xmax = fmaxf(xmax, xs[i]); }
  float ga = -1000.f, gb = -1000.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float xa = (fabsf(ys[i] - ymin) > 0.1f) ? -1000.f : xs[i];
    const float yb = (fabsf(xs[i] - xmax) > 0.1f) ? -1000.f : ys[i];
    ga = (i == 0) ? xa : fmaxf(ga, xa);
    gb = (i == 0) ? yb : fmaxf(gb, yb);
  }
  const float pcx = (an[0] + an[3]) * 0.5f, pcy = (an[1] + an[4]) * 0.5f, pcz = (an[2] + an[5]) * 0.5f;
  const float pw = an[3] - an[0], ph = an[4] - an[1], pd = an[5] - an[2];
  out[0] = (gx - pcx) / pw; out[1] = (gy - pcy) / ph; out[2] = (gt[2] - pcz) / pd;
  out[3] = logf(gw / pw); out[4] = logf(gh / ph); out[5] = logf(gt[5] / pd);
  out[6] = (ga - gx) / gw; out[7] = (gb - gy) / gh;
}

__device__ __forceinline__ void obb_to_aabb(const float *o, float *out) {
  const float cs = cosf(o[6]), sn = sinf(o[6]);
  const float bx = fabsf(o[3] / 2 * cs) + fabsf(o[4] / 2 * sn), by = fabsf(o[3] / 2 * sn) + fabsf(o[4] / 2 * cs);
  const float bz = o[5] / 2;
  out[0] = o[0] - bx; out[1] = o[1] - by; out[2] = o[2] - bz;
  out[3] = o[0] + bx; out[4] = o[1] + by; out[5] = o[2] + bz;
}

}  // namespace geo

// Differentiable rotated 3D IoU / GIoU / DIoU regression losses, forward AND gradient in one kernel (one lane per box pair).
//
// Replaces the ~40-kernel torch chain + sort op of reference nerf_rpn/model/rotated_iou/oriented_iou_loss.py:37-148,
// box_intersection_2d.py:11-176, min_enclosing_box.py:54-166 as used by RotatedIOULoss (model/rpn.py:133-164, fcos/loss.py:137-173)
// on the <= 128 sampled positives per scene.  The gradient with respect to the PREDICTED box (x,y,z,w,h,d,theta) is carried
// through the same arithmetic as forward-mode dual numbers (value + 7 partials), so every differentiable operation of the
// reference contributes exactly its autograd derivative, and the non-differentiable decisions -- validity masks, the vertex sort
// (sort_vert_kernel.cu:15-134), arg-min / arg-max selections, clamps -- are taken on the values, as autograd does.
// The target box is a constant (both RotatedIOULoss callers pass ground truth there).
// Compiled with -ffp-contract=off like the other geometry kernels (mask thresholds must see the written fp32 operations).
#include "geometry.cuh"

namespace {

constexpr int NG = 7;
struct D {
  float v;
  float g[NG];
};

__device__ __forceinline__ D cst(float v) {
  D r; r.v = v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = 0.f;
  return r;
}
__device__ __forceinline__ D var(float v, int k) { D r = cst(v); r.g[k] = 1.f; return r; }
__device__ __forceinline__ D operator+(const D &a, const D &b) { D r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = a.g[i] + b.g[i];
  return r; }
__device__ __forceinline__ D operator-(const D &a, const D &b) { D r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = a.g[i] - b.g[i];
  return r; }
__device__ __forceinline__ D operator-(const D &a) { D r; r.v = -a.v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = -a.g[i];
  return r; }
__device__ __forceinline__ D operator*(const D &a, const D &b) { D r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i];
  return r; }
__device__ __forceinline__ D operator/(const D &a, const D &b) { D r; r.v = a.v / b.v; const float inv = 1.f / b.v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) * inv;
  return r; }
__device__ __forceinline__ D operator+(const D &a, float b) { D r = a; r.v = a.v + b; return r; }
__device__ __forceinline__ D operator-(const D &a, float b) { D r = a; r.v = a.v - b; return r; }
__device__ __forceinline__ D operator-(float a, const D &b) { D r = -b; r.v = a - b.v; return r; }
__device__ __forceinline__ D operator*(const D &a, float b) { D r; r.v = a.v * b;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = a.g[i] * b;
  return r; }
__device__ __forceinline__ D operator*(float b, const D &a) { return a * b; }
__device__ __forceinline__ D dsqrt(const D &a) { D r; r.v = sqrtf(a.v); const float k = 0.5f / r.v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = a.g[i] * k;
  return r; }
__device__ __forceinline__ D dabs(const D &a) { return a.v < 0.f ? -a : a; }            // d|x|/dx = sign(x) (0 at 0: measure zero)
__device__ __forceinline__ D dmax(const D &a, const D &b) { return b.v > a.v ? b : a; } // first maximum wins (torch reductions)
__device__ __forceinline__ D dmin(const D &a, const D &b) { return b.v < a.v ? b : a; }
__device__ __forceinline__ D clamp0(const D &a) { return a.v < 0.f ? cst(0.f) : a; }    // clamp_min(0): zero gradient when clamped
__device__ __forceinline__ D dlog(const D &a) { D r; r.v = logf(a.v); const float inv = 1.f / a.v;
#pragma unroll
  for (int i = 0; i < NG; ++i) r.g[i] = a.g[i] * inv;
  return r; }

// box2corners_th (oriented_iou_loss.py:6-35): X = lx*c - ly*s + x, Y = lx*s + ly*c + y
__device__ __forceinline__ void corners(const D &x, const D &y, const D &w, const D &h, const D &a, D *X, D *Y) {
  D s, c;
  s.v = sinf(a.v); c.v = cosf(a.v);
#pragma unroll
  for (int i = 0; i < NG; ++i) { s.g[i] = c.v * a.g[i]; c.g[i] = -s.v * a.g[i]; }
  const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f};
  const float sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const D lx = w * sx[i], ly = h * sy[i];
    X[i] = lx * c - ly * s + x;
    Y[i] = lx * s + ly * c + y;
  }
}

// corners P inside box Q on VALUES (box_intersection_2d.py:54-79)
__device__ __forceinline__ void inside4(const float *PX, const float *PY, const float *QX, const float *QY, bool *in) {
  const float abx = QX[1] - QX[0], aby = QY[1] - QY[0];
  const float adx = QX[3] - QX[0], ady = QY[3] - QY[0];
  const float nab = abx * abx + aby * aby, nad = adx * adx + ady * ady;
  const float hi = (float)(1 + 1e-6), lo = -1e-6f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float amx = PX[i] - QX[0], amy = PY[i] - QY[0];
    const float r1 = (abx * amx + aby * amy) / nab;
    const float r2 = (adx * amx + ady * amy) / nad;
    in[i] = (r1 > lo) && (r1 < hi) && (r2 > lo) && (r2 < hi);
  }
}

struct PairOut {
  D iou, u3, zr;        // 3D IoU, union volume, z range of the enclosing box
  D X[8], Y[8];         // corners of box 1 (0-3, differentiable) and box 2 (4-7, constants)
};

// cal_iou_3d(verbose=True): oriented_iou_loss.py:82-107 + oriented_box_intersection_2d
__device__ void iou3d_dual(const float *pv, const float *q, PairOut &o) {
  D p[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) p[k] = var(pv[k], k);
  const D zt1 = p[2] + p[5] * 0.5f, zb1 = p[2] - p[5] * 0.5f;
  const float zt2 = q[2] + q[5] * 0.5f, zb2 = q[2] - q[5] * 0.5f;
  const D zov = clamp0(dmin(zt1, cst(zt2)) - dmax(zb1, cst(zb2)));
  corners(p[0], p[1], p[3], p[4], p[6], o.X, o.Y);
  {
    float BX[4], BY[4];
    geo::corners2d(q[0], q[1], q[3], q[4], q[6], BX, BY);
#pragma unroll
    for (int i = 0; i < 4; ++i) { o.X[4 + i] = cst(BX[i]); o.Y[4 + i] = cst(BY[i]); }
  }
  // ---- 24 candidate vertices: 4 + 4 corners, 16 edge intersections
  D vx[24], vy[24];
  bool ok[24];
  float AXv[4], AYv[4], BXv[4], BYv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { AXv[i] = o.X[i].v; AYv[i] = o.Y[i].v; BXv[i] = o.X[4 + i].v; BYv[i] = o.Y[4 + i].v; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { vx[i] = o.X[i]; vy[i] = o.Y[i]; }
  inside4(AXv, AYv, BXv, BYv, ok);
  inside4(BXv, BYv, AXv, AYv, ok + 4);
  for (int i = 0; i < 4; ++i) {
    const D x1 = o.X[i], y1 = o.Y[i], x2 = o.X[(i + 1) & 3], y2 = o.Y[(i + 1) & 3];
    for (int j = 0; j < 4; ++j) {
      const float x3 = BXv[j], y3 = BYv[j], x4 = BXv[(j + 1) & 3], y4 = BYv[(j + 1) & 3];
      const D num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
      const D den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
      const D den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3);
      bool v = false;
      if (num.v != 0.0f) {
        const float t = den_t.v / num.v, u = -den_u.v / num.v;
        v = (t > 0.f) && (t < 1.f) && (u > 0.f) && (u < 1.f);
      }
      const int k = 8 + i * 4 + j;
      ok[k] = v;
      if (v) {
        const D ts = den_t / (num + 1e-8f);
        vx[k] = x1 + ts * (x2 - x1);
        vy[k] = y1 + ts * (y2 - y1);
      } else {
        vx[k] = cst(0.f); vy[k] = cst(0.f);
      }
    }
  }
  // ---- vertex order on the values (nrpn_sort_vertices_f32 == sort_vert_kernel.cu:42-134)
  int nv = 0;
  float mx = 0.f, my = 0.f;
  for (int k = 0; k < 24; ++k) { nv += ok[k] ? 1 : 0; mx += ok[k] ? vx[k].v : 0.f; my += ok[k] ? vy[k].v : 0.f; }
  int pad = 23;
  for (int j = 8; j < 24; ++j)
    if (!ok[j]) { pad = j; break; }
  int loc[9];
  if (nv < 3) {
    for (int j = 0; j < 9; ++j) loc[j] = pad;
  } else {
    mx /= (float)nv; my /= (float)nv;
    int n = nv > 8 ? 8 : nv;
    float px = 0.f, py = 0.f;
    for (int j = 0; j < n; ++j) {
      float bx = 1.0f, by = (float)(-1e-8);
      int take = 0;
      for (int k = 0; k < 24; ++k) {
        if (!ok[k]) continue;
        const float x = vx[k].v - mx, y = vy[k].v - my;
        bool c = geo::vert_before(x, y, bx, by);
        if (j > 0) c = c && geo::vert_before(px, py, x, y);
        if (c) { bx = x; by = y; take = k; }
      }
      loc[j] = take;
      px = vx[take].v - mx; py = vy[take].v - my;
    }
    loc[n] = loc[0];
    for (int j = n + 1; j < 9; ++j) loc[j] = pad;
    if (n == 8) {
      int dup = 0;
      for (int j = 0; j < 4; ++j)
        for (int k = 4; k < 8; ++k) dup += (loc[k] == loc[j]);
      if (dup == 4) {
        loc[4] = loc[0];
        for (int j = 5; j < 9; ++j) loc[j] = pad;
      }
    }
  }
  // ---- shoelace over the 9 gathered vertices
  D total = cst(0.f);
  for (int m = 0; m < 8; ++m) {
    const int a = loc[m], b = loc[m + 1];
    total = total + (vx[a] * vy[b] - vy[a] * vx[b]);
  }
  const D inter2 = dabs(total) * 0.5f;
  const float a2 = q[3] * q[4];
  const D u2 = p[3] * p[4] + a2 - inter2;
  const D iou2 = inter2 / u2;
  const D inter3 = iou2 * u2 * zov;
  const float v2 = q[3] * q[4] * q[5];
  o.u3 = p[3] * p[4] * p[5] + v2 - inter3;
  o.iou = inter3 / o.u3;
  o.zr = clamp0(dmax(zt1, cst(zt2)) - dmin(zb1, cst(zb2)));
}

// smallest_bounding_box (min_enclosing_box.py:54-125): over the 24 candidate edges through two of the 8 corners
__device__ void enclosing_wh(const D *X, const D *Y, D &w, D &h) {
  float best = 0.f;
  bool have = false;
  for (int i = 0; i < 8; ++i) {
    for (int j = i + 1; j < 8; ++j) {
      if ((i == 0 && j == 2) || (i == 1 && j == 3) || (i == 5 && j == 7) || (i == 4 && j == 6)) continue;
      const D x1 = X[i], y1 = Y[i], x2 = X[j], y2 = Y[j];
      const D k = (y2 - y1) / (x2 - x1 + 1e-8f);
      const D nrm = dsqrt(cst(1.f) + k * k);
      // projections of all 8 points on the edge direction, in the order [edge point 1, edge point 2, the other six]
      D pmax, pmin, dmx, dmn, amax;
      bool fp = true, fd = true;
      const D dx = x2 - x1, dy = y2 - y1;
      const D den = dsqrt(dy * dy + dx * dx + 1e-14f);
      int order[8];
      order[0] = i; order[1] = j;
      int c = 2;
      for (int t = 0; t < 8; ++t)
        if (t != i && t != j) order[c++] = t;
      for (int t = 0; t < 8; ++t) {
        const int pt = order[t];
        const D pr = (X[pt] + Y[pt] * k) / nrm;
        if (fp) { pmax = pr; pmin = pr; fp = false; } else { pmax = dmax(pmax, pr); pmin = dmin(pmin, pr); }
        if (t >= 2) {
          const D dd = (dy * X[pt] - dx * Y[pt] + x2 * y1 - y2 * x1) / den;
          const D ad = dabs(dd);
          if (fd) { dmx = dd; dmn = dd; amax = ad; fd = false; } else { dmx = dmax(dmx, dd); dmn = dmin(dmn, dd); amax = dmax(amax, ad); }
        }
      }
      const D prange = pmax - pmin;
      const D dr1 = dmx - dmn;
      const D drange = amax.v > dr1.v ? amax : dr1;            // torch.max(a, b): the larger operand carries the gradient
      float area = prange.v * drange.v;
      if (area == 0.f) area += 1e8f;
      if (!have || area < best) { best = area; have = true; w = prange; h = drange; }
    }
  }
}

// mode: 0 = 'iou'  -log((iou*u+1)/(u+1)),  1 = 'linear_iou'  1 - (iou*u+1)/(u+1),  2 = 'giou',  3 = 'diou'
__global__ void rotated_iou_loss_kernel(const float *__restrict__ pred, const float *__restrict__ target, int64_t n, int mode,
                                        float *__restrict__ loss, float *__restrict__ grad, float *__restrict__ iou_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[7], q[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { p[k] = pred[i * 7 + k]; q[k] = target[i * 7 + k]; }
  PairOut o;
  iou3d_dual(p, q, o);
  D l;
  if (mode == 0 || mode == 1) {
    const D r = (o.iou * o.u3 + 1.0f) / (o.u3 + 1.0f);
    l = (mode == 0) ? -dlog(r) : (1.0f - r);
  } else {
    D w, h;
    enclosing_wh(o.X, o.Y, w, h);
    if (mode == 2) {
      const D vc = o.zr * w * h;
      l = 1.0f - o.iou + (vc - o.u3) / vc;
    } else {
      D c[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) c[k] = var(p[k], k) - q[k];
      const D d2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
      l = 1.0f - o.iou + d2 / (w * w + h * h + o.zr * o.zr);
    }
  }
  loss[i] = l.v;
  if (iou_out) iou_out[i] = o.iou.v;
#pragma unroll
  for (int k = 0; k < 7; ++k) grad[i * 7 + k] = l.g[k];
}

// ---------------------------------------------------------------------------------------------------------------------
// 2-D projection smooth-L1 of the RPN (reference model/rpn.py:37-102 get_w2cs / project / obb2points_3d, 421-453): the two extreme
// points of every predicted and matched ground-truth box are projected into the four fixed views (camera = M @ [x y z 1]^T,
// picture = K @ camera[:3], (u, v) = picture[:2] / picture[2]) and compared with smooth-L1, summed, / n / max_mesh_dim.  VALUE
// only (the reported loss_rpn_box_reg_2d when its weight is 0, the reference default): one workgroup, fixed summation order.
// The torch formulation -- ~45 launches for <= 128 boxes -- remains the path when the term is trained through.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void box_point(const float *b, int box_dim, int e, float &x, float &y, float &z) {
  if (box_dim == 6) {   // AABB: min corner, max corner
    x = b[3 * e]; y = b[3 * e + 1]; z = b[3 * e + 2];
    return;
  }
  const float c = cosf(b[6]), s = sinf(b[6]);
  const float vx = b[3] / 2 * c - b[4] / 2 * s, vy = b[3] / 2 * s + b[4] / 2 * c, vz = b[5] / 2;
  const float sg = e ? 1.f : -1.f;
  x = b[0] + sg * vx; y = b[1] + sg * vy; z = b[2] + sg * vz;
}

__device__ __forceinline__ void view_uv(const float *M, const float *K, float x, float y, float z, float &u, float &v) {
  float cam[3], pic[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) cam[r] = M[r * 4] * x + M[r * 4 + 1] * y + M[r * 4 + 2] * z + M[r * 4 + 3];
#pragma unroll
  for (int r = 0; r < 3; ++r) pic[r] = K[r * 3] * cam[0] + K[r * 3 + 1] * cam[1] + K[r * 3 + 2] * cam[2];
  u = pic[0] / pic[2];
  v = pic[1] / pic[2];
}

__device__ __forceinline__ float smooth_l1(float d, float beta) {
  const float a = fabsf(d);
  return a < beta ? 0.5f * d * d / beta : a - 0.5f * beta;
}

__global__ void __launch_bounds__(256) projection_loss_kernel(const float *__restrict__ pred, const float *__restrict__ target, long long n, int box_dim,
                                                              const float *__restrict__ views, const float *__restrict__ intr, float beta,
                                                              float max_mesh_dim, float *__restrict__ out) {
  __shared__ float red[256];
  __shared__ float Ms[64], Ks[9];
  if (threadIdx.x < 64) Ms[threadIdx.x] = views[threadIdx.x];
  if (threadIdx.x < 9) Ks[threadIdx.x] = intr[threadIdx.x];
  __syncthreads();
  float acc = 0.f;
  for (long long item = threadIdx.x; item < 2 * n; item += 256) {
    const long long b = item >> 1;
    const int e = (int)(item & 1);
    float px, py, pz, tx, ty, tz;
    box_point(pred + b * box_dim, box_dim, e, px, py, pz);
    box_point(target + b * box_dim, box_dim, e, tx, ty, tz);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float pu, pv, tu, tv;
      view_uv(Ms + v * 16, Ks, px, py, pz, pu, pv);
      view_uv(Ms + v * 16, Ks, tx, ty, tz, tu, tv);
      acc += smooth_l1(pu - tu, beta) + smooth_l1(pv - tv, beta);
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (float)n / max_mesh_dim;
}

}  // namespace

extern "C" int nrpn_projection_loss_f32(const float *pred, const float *target, int64_t n, int box_dim, const float *views, const float *intrinsics,
                                        float beta, float max_mesh_dim, float *out, nrpn_stream_t stream) {
  NRPN_REQUIRE(n >= 0 && (box_dim == 6 || box_dim == 7) && beta > 0.f && max_mesh_dim > 0.f, "projection_loss: bad arguments (n=%lld box_dim=%d)",
               (long long)n, box_dim);
  NRPN_REQUIRE(views && intrinsics && out && (n == 0 || (pred && target)), "projection_loss: null pointer");
  hipLaunchKernelGGL(projection_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), pred, target, (long long)n, box_dim, views, intrinsics, beta,
                     max_mesh_dim, out);
  NRPN_LAUNCH_CHECK("projection_loss");
  return NRPN_OK;
}

extern "C" int nrpn_rotated_iou_loss_f32(const float *pred, const float *target, int64_t n, int mode, float *loss, float *grad, float *iou,
                                         nrpn_stream_t stream) {
  NRPN_REQUIRE(n >= 0 && mode >= 0 && mode <= 3, "rotated_iou_loss: bad arguments (n=%lld mode=%d)", (long long)n, mode);
  if (n == 0) return NRPN_OK;
  NRPN_REQUIRE(pred && target && loss && grad, "rotated_iou_loss: null pointer");
  hipLaunchKernelGGL(rotated_iou_loss_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(64), 0, as_stream(stream), pred, target, n, mode, loss, grad,
                     iou);
  NRPN_LAUNCH_CHECK("rotated_iou_loss");
  return NRPN_OK;
}

// ROIPool WITHOUT the RoIAlign op -- the reference CLI's default second-stage pooling (nerf_rpn/model/detector.py, ``ROIPool(use_cuda=False)``):
//   * axis-aligned RoIs (:397-438 normal_forward): the integer crop [floor(lo / s), floor(hi / s)] of the level's map (python slicing: the high
//     end is clipped to the map), zero-padded at the high side to a multiple of the output size, max-pool with kernel = stride =
//     ceil(extent / output) (:377-384, :425-432);
//   * rotated RoIs (:264-395): a regular grid of ceil(extent / s) points per axis rotated by theta about the box centre, every point the
//     reference's 8-corner blend  sum_corners feat[corner] * (1 - |dx| |dy| |dz|) / 8  (zero outside the map; NOT a trilinear interpolation --
//     kept as it is, RCNN weights were trained on it), then the same adaptive max-pool ('pooling') or a trilinear resize with aligned corners
//     ('interpolation', :385-393).
// Round 4 ran these as per-RoI torch loops (F.max_pool3d / F.interpolate on crops); here each is ONE launch per pyramid level over all RoIs of a
// scene, forward and backward, on channels-last maps [X][Y][Z][C] (f32 or bf16), pooled rows fp32 [R][o0][o1][o2][C] like the reference's
// ``.float()``.  A block = one (RoI, output voxel); its lanes = channels (coalesced rows).  RoIs of other levels exit at once (level id per RoI):
// no host-side grouping, no synchronisation.  Max ties go to the FIRST element in scan order (x, y, z ascending; padding zeros take part where
// they lie), as torch's max_pool3d does.  Backward scatters through 64-bit fixed-point integer atomics (2^44, as csrc/roialign.hip): integer
// addition is associative, so the gradient is bit-identical from run to run.
// Compiled with -ffp-contract=off: floor / ceil of the rotated sample positions decide which voxels are read.
#include "common.h"

namespace {
typedef unsigned short bf16s;
constexpr double kFix = 17592186044416.0;     // 2^44
constexpr float kFixClamp = 262144.f;         // 2^18

__device__ __forceinline__ void fix_add(long long *dst, float g) {
  g = fminf(fmaxf(g, -kFixClamp), kFixClamp);
  if (g != 0.f) atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)__double2ll_rn((double)g * kFix));
}

struct PoolGeom { int o0, o1, o2; };

// ---------------------------------------------------------------------------------------------------------------------
// axis-aligned crops.  crop [R][6] = (start x, y, z, size x, y, z) in voxels of the RoI's level (sizes <= 0: the RoI pools to zeros)
// MODE 0: forward (out, arg);  MODE 1: backward (dout -> ws at arg)
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ void roipool_aabb_kernel(const T *__restrict__ feat, int Y, int Z, int C, const int *__restrict__ crop, const int *__restrict__ level_of,
                                    int level, PoolGeom g, float *__restrict__ out, int *__restrict__ arg, const float *__restrict__ dout,
                                    long long *__restrict__ ws) {
  const int bins = g.o0 * g.o1 * g.o2;
  const long long r = blockIdx.x / bins;
  if (level_of[r] != level) return;
  const int w = blockIdx.x % bins;
  const int w2 = w % g.o2, w1 = (w / g.o2) % g.o1, w0 = w / (g.o2 * g.o1);
  const int *cr = crop + r * 6;
  const int s0 = cr[3], s1 = cr[4], s2 = cr[5];
  const long long orow = ((long long)r * bins + w) * C;
  if (MODE == 1) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const int a = arg[orow + c];
      if (a >= 0) fix_add(ws + (long long)a * C + c, dout[orow + c]);
    }
    return;
  }
  if (s0 <= 0 || s1 <= 0 || s2 <= 0) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) { out[orow + c] = 0.f; arg[orow + c] = -1; }
    return;
  }
  const int k0 = (s0 + g.o0 - 1) / g.o0, k1 = (s1 + g.o1 - 1) / g.o1, k2 = (s2 + g.o2 - 1) / g.o2;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float best = 0.f;
    int bi = -1;
    bool first = true;
    for (int a = w0 * k0; a < (w0 + 1) * k0; ++a)
      for (int b = w1 * k1; b < (w1 + 1) * k1; ++b)
        for (int d = w2 * k2; d < (w2 + 1) * k2; ++d) {
          const bool real = a < s0 && b < s1 && d < s2;
          const int vox = real ? ((cr[0] + a) * Y + cr[1] + b) * Z + cr[2] + d : -1;
          const float v = real ? elem<T>::ld(feat + (long long)vox * C + c) : 0.f;        // zero padding at the high side takes part in the max
          if (first || v > best) { best = v; bi = vox; first = false; }
        }
    out[orow + c] = best;
    arg[orow + c] = bi;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// rotated RoIs.  rois [R][7] = (x, y, z, w, l, h, theta) in input voxels, extents already enlarged; s = input voxels per voxel of this level
// ---------------------------------------------------------------------------------------------------------------------
struct Obb {
  int g0, g1, g2;
  float cx, cy, cz, cs, sn;
};
__device__ __forceinline__ Obb obb_setup(const float *r, float s) {
  Obb o;
  o.g0 = max(1, (int)ceilf(r[3] / s)); o.g1 = max(1, (int)ceilf(r[4] / s)); o.g2 = max(1, (int)ceilf(r[5] / s));
  o.cx = r[0] / s; o.cy = r[1] / s; o.cz = r[2] / s;
  o.cs = cosf(r[6]); o.sn = sinf(r[6]);
  return o;
}
// position of grid point (i, j, k): rot(theta) @ (index - (g - 1) / 2) + centre / s   (detector.py:306-315)
__device__ __forceinline__ void obb_point(const Obb &o, int i, int j, int k, float &x, float &y, float &z) {
  const float lx = (float)i - ((float)o.g0 - 1.f) / 2.f, ly = (float)j - ((float)o.g1 - 1.f) / 2.f, lz = (float)k - ((float)o.g2 - 1.f) / 2.f;
  x = (o.cs * lx + (-o.sn) * ly) + o.cx;
  y = (o.sn * lx + o.cs * ly) + o.cy;
  z = lz + o.cz;
}
struct Corners {
  int vox[8];
  float wt[8];       // (1 - |dx| |dy| |dz|) per corner; the blend is sum(feat * wt) * inside / 8
  bool inside;
};
__device__ __forceinline__ Corners obb_corners(float x, float y, float z, int X, int Y, int Z) {
  Corners q;
  q.inside = x >= 0.f && x <= (float)(X - 1) && y >= 0.f && y <= (float)(Y - 1) && z >= 0.f && z <= (float)(Z - 1);
  const float qx[2] = {floorf(x), ceilf(x)}, qy[2] = {floorf(y), ceilf(y)}, qz[2] = {floorf(z), ceilf(z)};
#pragma unroll
  for (int n = 0; n < 8; ++n) {        // the reference's order: (floor|ceil) of x outermost, z innermost
    const int a = n >> 2, b = (n >> 1) & 1, d = n & 1;
    const int ix = min(max((int)qx[a], 0), X - 1), iy = min(max((int)qy[b], 0), Y - 1), iz = min(max((int)qz[d], 0), Z - 1);
    q.vox[n] = (ix * Y + iy) * Z + iz;
    q.wt[n] = 1.f - fabsf(x - qx[a]) * fabsf(y - qy[b]) * fabsf(z - qz[d]);
  }
  return q;
}
template <typename T>
__device__ __forceinline__ float obb_blend(const T *__restrict__ feat, const Corners &q, int C, int c) {
  if (!q.inside) return 0.f;          // (the reference multiplies by the 0 / 1 mask: the same value, -0.0 aside)
  float acc = 0.f;
#pragma unroll
  for (int n = 0; n < 8; ++n) acc = acc + elem<T>::ld(feat + (long long)q.vox[n] * C + c) * q.wt[n];
  return acc / 8.f;
}
__device__ __forceinline__ void obb_scatter(long long *__restrict__ ws, const Corners &q, int C, int c, float g) {
  if (!q.inside || g == 0.f) return;
#pragma unroll
  for (int n = 0; n < 8; ++n) fix_add(ws + (long long)q.vox[n] * C + c, g * q.wt[n] / 8.f);
}
// torch's aligned-corner linear resize of an axis: source coordinate, lower tap, weights (upsample_trilinear3d, align_corners=True)
__device__ __forceinline__ void lin_tap(int dst, int in, int out, int &i0, int &i1, float &l0, float &l1) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)dst;
  i0 = min((int)src, in - 1);
  i1 = min(i0 + 1, in - 1);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

// KIND 0: adaptive max-pool of the sampled grid ('pooling'); KIND 1: trilinear resize ('interpolation').  MODE 0 forward, 1 backward.
template <typename T, int KIND, int MODE>
__global__ void roipool_obb_kernel(const T *__restrict__ feat, int X, int Y, int Z, int C, const float *__restrict__ rois,
                                   const int *__restrict__ level_of, int level, float s, PoolGeom g, float *__restrict__ out,
                                   int *__restrict__ arg, const float *__restrict__ dout, long long *__restrict__ ws) {
  const int bins = g.o0 * g.o1 * g.o2;
  const long long r = blockIdx.x / bins;
  if (level_of[r] != level) return;
  const int w = blockIdx.x % bins;
  const int w2 = w % g.o2, w1 = (w / g.o2) % g.o1, w0 = w / (g.o2 * g.o1);
  const Obb o = obb_setup(rois + r * 7, s);
  const long long orow = ((long long)r * bins + w) * C;
  if (KIND == 0) {
    const int k0 = (o.g0 + g.o0 - 1) / g.o0, k1 = (o.g1 + g.o1 - 1) / g.o1, k2 = (o.g2 + g.o2 - 1) / g.o2;
    if (MODE == 1) {
      for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const int a = arg[orow + c];
        if (a < 0) continue;
        const int k = a % o.g2, j = (a / o.g2) % o.g1, i = a / (o.g2 * o.g1);
        float x, y, z;
        obb_point(o, i, j, k, x, y, z);
        obb_scatter(ws, obb_corners(x, y, z, X, Y, Z), C, c, dout[orow + c]);
      }
      return;
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float best = 0.f;
      int bi = -1;
      bool first = true;
      for (int a = w0 * k0; a < (w0 + 1) * k0; ++a)
        for (int b = w1 * k1; b < (w1 + 1) * k1; ++b)
          for (int d = w2 * k2; d < (w2 + 1) * k2; ++d) {
            const bool real = a < o.g0 && b < o.g1 && d < o.g2;
            float v = 0.f;
            if (real) {
              float x, y, z;
              obb_point(o, a, b, d, x, y, z);
              v = obb_blend(feat, obb_corners(x, y, z, X, Y, Z), C, c);
            }
            if (first || v > best) { best = v; bi = real ? (a * o.g1 + b) * o.g2 + d : -1; first = false; }
          }
      out[orow + c] = best;
      arg[orow + c] = bi;
    }
  } else {
    int i0[3], i1[3];
    float l0[3], l1[3];
    lin_tap(w0, o.g0, g.o0, i0[0], i1[0], l0[0], l1[0]);
    lin_tap(w1, o.g1, g.o1, i0[1], i1[1], l0[1], l1[1]);
    lin_tap(w2, o.g2, g.o2, i0[2], i1[2], l0[2], l1[2]);
    Corners q[8];
    float tw[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
      const int a = n >> 2, b = (n >> 1) & 1, d = n & 1;
      float x, y, z;
      obb_point(o, a ? i1[0] : i0[0], b ? i1[1] : i0[1], d ? i1[2] : i0[2], x, y, z);
      q[n] = obb_corners(x, y, z, X, Y, Z);
      tw[n] = (a ? l1[0] : l0[0]) * ((b ? l1[1] : l0[1]) * (d ? l1[2] : l0[2]));
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      if (MODE == 0) {
        // torch: t0 * (h0 * (w0 * v000 + w1 * v001) + h1 * (w0 * v010 + w1 * v011)) + t1 * (...)
        float v[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) v[n] = obb_blend(feat, q[n], C, c);
        const float lo = l0[1] * (l0[2] * v[0] + l1[2] * v[1]) + l1[1] * (l0[2] * v[2] + l1[2] * v[3]);
        const float hi = l0[1] * (l0[2] * v[4] + l1[2] * v[5]) + l1[1] * (l0[2] * v[6] + l1[2] * v[7]);
        out[orow + c] = l0[0] * lo + l1[0] * hi;
      } else {
        const float gout = dout[orow + c];
#pragma unroll
        for (int n = 0; n < 8; ++n) obb_scatter(ws, q[n], C, c, gout * tw[n]);
      }
    }
  }
}

template <typename T>
__global__ void roipool_fixed_to_float_kernel(const long long *__restrict__ ws, T *__restrict__ dst, long long count) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
    elem<T>::st(dst + i, (float)((double)ws[i] * (1.0 / kFix)));
}

int check(const char *who, int num_rois, int x, int y, int z, int c, int o0, int o1, int o2, int dtype) {
  if (!(num_rois >= 0 && x > 0 && y > 0 && z > 0 && c > 0 && o0 > 0 && o1 > 0 && o2 > 0)) return nrpn_fail(NRPN_ERR_ARG, "%s: bad sizes", who);
  if (!(dtype == NRPN_F32 || dtype == NRPN_BF16)) return nrpn_fail(NRPN_ERR_ARG, "%s: bad dtype %d", who, dtype);
  if (!((long long)num_rois * o0 * o1 * o2 < (1ll << 31) && (long long)x * y * z < (1ll << 31)))
    return nrpn_fail(NRPN_ERR_ARG, "%s: too many bins / voxels", who);
  return 0;
}
static inline int lanes_for(int c) { return c >= 256 ? 256 : (c >= 128 ? 128 : 64); }
}  // namespace

extern "C" size_t nrpn_roipool_bwd_workspace_bytes(int x, int y, int z, int c) { return (size_t)x * y * z * c * 8; }

extern "C" int nrpn_roipool_aabb_fwd(const void *feat, int x, int y, int z, int c, const int32_t *crop, const int32_t *level_of, int level, int num_rois,
                                     int o0, int o1, int o2, float *out, int32_t *argmax, int dtype, nrpn_stream_t stream) {
  if (int rc = check("roipool_aabb_fwd", num_rois, x, y, z, c, o0, o1, o2, dtype)) return rc;
  if (num_rois == 0) return NRPN_OK;
  NRPN_REQUIRE(feat && crop && level_of && out && argmax, "roipool_aabb_fwd: null pointer");
  const dim3 grid((unsigned)(num_rois * o0 * o1 * o2));
  const PoolGeom g{o0, o1, o2};
  if (dtype == NRPN_F32)
    hipLaunchKernelGGL((roipool_aabb_kernel<float, 0>), grid, dim3(lanes_for(c)), 0, as_stream(stream), (const float *)feat, y, z, c, crop, level_of, level, g,
                       out, argmax, (const float *)nullptr, (long long *)nullptr);
  else
    hipLaunchKernelGGL((roipool_aabb_kernel<bf16s, 0>), grid, dim3(lanes_for(c)), 0, as_stream(stream), (const bf16s *)feat, y, z, c, crop, level_of, level, g,
                       out, argmax, (const float *)nullptr, (long long *)nullptr);
  NRPN_LAUNCH_CHECK("roipool_aabb_fwd");
  return NRPN_OK;
}

extern "C" int nrpn_roipool_aabb_bwd(const float *dout, const int32_t *argmax, const int32_t *crop, const int32_t *level_of, int level, int num_rois,
                                     int x, int y, int z, int c, int o0, int o1, int o2, void *dfeat, void *workspace, int dtype, nrpn_stream_t stream) {
  if (int rc = check("roipool_aabb_bwd", num_rois, x, y, z, c, o0, o1, o2, dtype)) return rc;
  NRPN_REQUIRE(dfeat && workspace && (num_rois == 0 || (dout && argmax && crop && level_of)), "roipool_aabb_bwd: null pointer");
  hipStream_t st = as_stream(stream);
  const long long count = (long long)x * y * z * c;
  NRPN_HIP(hipMemsetAsync(workspace, 0, (size_t)count * 8, st));
  if (num_rois > 0) {
    const PoolGeom g{o0, o1, o2};
    hipLaunchKernelGGL((roipool_aabb_kernel<float, 1>), dim3((unsigned)(num_rois * o0 * o1 * o2)), dim3(lanes_for(c)), 0, st, (const float *)nullptr, y, z, c,
                       crop, level_of, level, g, (float *)nullptr, const_cast<int32_t *>(argmax), dout, (long long *)workspace);
  }
  const int blocks = (int)min((long long)8192, (count + 255) / 256);
  if (dtype == NRPN_F32) hipLaunchKernelGGL(roipool_fixed_to_float_kernel<float>, dim3(blocks), dim3(256), 0, st, (const long long *)workspace, (float *)dfeat, count);
  else hipLaunchKernelGGL(roipool_fixed_to_float_kernel<bf16s>, dim3(blocks), dim3(256), 0, st, (const long long *)workspace, (bf16s *)dfeat, count);
  NRPN_LAUNCH_CHECK("roipool_aabb_bwd");
  return NRPN_OK;
}

#define NRPN_OBB_LAUNCH(T_, KIND_, MODE_, feat_, out_, arg_, dout_, ws_)                                                                                 \
  hipLaunchKernelGGL((roipool_obb_kernel<T_, KIND_, MODE_>), grid, dim3(lanes_for(c)), 0, st, (const T_ *)(feat_), x, y, z, c, rois, level_of, level, scale, \
                     g, out_, arg_, dout_, ws_)

extern "C" int nrpn_roipool_obb_fwd(const void *feat, int x, int y, int z, int c, const float *rois, const int32_t *level_of, int level, int num_rois,
                                    float scale, int interpolation, int o0, int o1, int o2, float *out, int32_t *argmax, int dtype, nrpn_stream_t stream) {
  if (int rc = check("roipool_obb_fwd", num_rois, x, y, z, c, o0, o1, o2, dtype)) return rc;
  if (num_rois == 0) return NRPN_OK;
  NRPN_REQUIRE(feat && rois && level_of && out && (interpolation || argmax) && scale > 0.f, "roipool_obb_fwd: null pointer / bad scale");
  hipStream_t st = as_stream(stream);
  const dim3 grid((unsigned)(num_rois * o0 * o1 * o2));
  const PoolGeom g{o0, o1, o2};
  if (dtype == NRPN_F32) {
    if (interpolation) NRPN_OBB_LAUNCH(float, 1, 0, feat, out, argmax, (const float *)nullptr, (long long *)nullptr);
    else NRPN_OBB_LAUNCH(float, 0, 0, feat, out, argmax, (const float *)nullptr, (long long *)nullptr);
  } else {
    if (interpolation) NRPN_OBB_LAUNCH(bf16s, 1, 0, feat, out, argmax, (const float *)nullptr, (long long *)nullptr);
    else NRPN_OBB_LAUNCH(bf16s, 0, 0, feat, out, argmax, (const float *)nullptr, (long long *)nullptr);
  }
  NRPN_LAUNCH_CHECK("roipool_obb_fwd");
  return NRPN_OK;
}

extern "C" int nrpn_roipool_obb_bwd(const float *dout, const int32_t *argmax, const float *rois, const int32_t *level_of, int level, int num_rois,
                                    float scale, int interpolation, int x, int y, int z, int c, int o0, int o1, int o2, void *dfeat, void *workspace,
                                    int dtype, nrpn_stream_t stream) {
  if (int rc = check("roipool_obb_bwd", num_rois, x, y, z, c, o0, o1, o2, dtype)) return rc;
  NRPN_REQUIRE(dfeat && workspace && (num_rois == 0 || (dout && rois && level_of && (interpolation || argmax))) && scale > 0.f,
               "roipool_obb_bwd: null pointer / bad scale");
  hipStream_t st = as_stream(stream);
  const long long count = (long long)x * y * z * c;
  NRPN_HIP(hipMemsetAsync(workspace, 0, (size_t)count * 8, st));
  if (num_rois > 0) {
    const dim3 grid((unsigned)(num_rois * o0 * o1 * o2));
    const PoolGeom g{o0, o1, o2};
    float *no_out = nullptr;
    int32_t *arg = const_cast<int32_t *>(argmax);
    long long *ws = (long long *)workspace;
    const void *no_feat = nullptr;
    if (interpolation) NRPN_OBB_LAUNCH(float, 1, 1, no_feat, no_out, arg, dout, ws);
    else NRPN_OBB_LAUNCH(float, 0, 1, no_feat, no_out, arg, dout, ws);
  }
  const int blocks = (int)min((long long)8192, (count + 255) / 256);
  if (dtype == NRPN_F32) hipLaunchKernelGGL(roipool_fixed_to_float_kernel<float>, dim3(blocks), dim3(256), 0, st, (const long long *)workspace, (float *)dfeat, count);
  else hipLaunchKernelGGL(roipool_fixed_to_float_kernel<bf16s>, dim3(blocks), dim3(256), 0, st, (const long long *)workspace, (bf16s *)dfeat, count);
  NRPN_LAUNCH_CHECK("roipool_obb_bwd");
  return NRPN_OK;
}
#undef NRPN_OBB_LAUNCH

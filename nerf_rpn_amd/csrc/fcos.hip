// FCOS variant of the hot path on gfx950 (reference nerf_rpn/model/fcos/{fcos.py,inference.py,loss.py,utils.py}):
//   * GroupNorm(32, C) (+ fused ReLU) forward / backward on channels-last activations (the tower norm, fcos.py:57,69),
//   * the head epilogue (per-level Scale, ReLU on the 6 distances, x stride at test; fcos.py:104-128) writing logits /
//     regressions / centerness straight into the flattened all-level order of the loss and the post-processor,
//   * target assignment with centre sampling (loss.py:209-437), the sigmoid focal loss with its gradient,
//   * score = sigmoid(cls) * sigmoid(ctr) with padding / threshold masks and the candidate decode (AABB clip or
//     midpoint-offset OBB decode; inference.py:56-140, utils.py:12-62).
// Locations are never materialised: (level, scene, voxel) -> (x, y, z) = idx * stride + stride / 2 is index arithmetic.
// Compiled with -ffp-contract=off: the targets compare fp32 expressions against thresholds exactly as torch evaluates them.
#include "common.h"
#include <atomic>
#include "geometry.cuh"

typedef unsigned short bf16s;

#define DISPATCH_T(dtype, ...)                                 \
  if ((dtype) == NRPN_F32) { typedef float T; __VA_ARGS__; }   \
  else { typedef bf16s T; __VA_ARGS__; }

static inline int ew_blocks(long long work, int cap = 8192) { long long b = (work + 255) / 256; return (int)(b > cap ? cap : (b < 1 ? 1 : b)); }

// =====================================================================================================================
// GroupNorm
// =====================================================================================================================
// 4 consecutive channels per thread (8/16-byte accesses); C % 4 == 0
template <typename T> struct gvec4;
template <> struct gvec4<float> {
  typedef __attribute__((ext_vector_type(4))) float f4;
  static __device__ __forceinline__ void ld(const float *p, float *v) { const f4 t = *reinterpret_cast<const f4 *>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
  static __device__ __forceinline__ void st(float *p, const float *v) { f4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4 *>(p) = t; }
};
template <> struct gvec4<bf16s> {
  typedef __attribute__((ext_vector_type(4))) unsigned short us4;
  static __device__ __forceinline__ void ld(const bf16s *p, float *v) {
    const us4 t = *reinterpret_cast<const us4 *>(p);
    v[0] = bf16_bits_to_f32(t[0]); v[1] = bf16_bits_to_f32(t[1]); v[2] = bf16_bits_to_f32(t[2]); v[3] = bf16_bits_to_f32(t[3]);
  }
  static __device__ __forceinline__ void st(bf16s *p, const float *v) {
    us4 t = {f32_to_bf16_bits(v[0]), f32_to_bf16_bits(v[1]), f32_to_bf16_bits(v[2]), f32_to_bf16_bits(v[3])};
    *reinterpret_cast<us4 *>(p) = t;
  }
};

// per-(sample, channel) sums over the sample's rows: fwd (sum x, sum x^2), bwd (sum g, sum g*x) with g = dy * relu mask.
// Block = one slab of rows of one sample; thread = (4-channel group, row lane), 8/16-byte loads, 4 rows in flight per thread;
// the row lanes are reduced through LDS and the block writes ONE partial row [C][2]; gn_reduce_kernel sums the partial rows
// into ws[n][c][2].  (The first version added every block's sums with fp32 atomics and read one element per thread.)
constexpr int kGnMaxParts = 1024;       // partial rows per sample
static inline int gn_slab(long long rows) { const long long s = (rows + kGnMaxParts - 1) / kGnMaxParts; return (int)(s < 32 ? 32 : s); }

template <typename T, bool BWD>
__global__ void __launch_bounds__(256) gn_partial_kernel(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy,
                                                         float *__restrict__ partial, long long rows, int c, int relu, int nparts, int slab) {
  __shared__ float red[2][256][4];
  const int n = blockIdx.y;
  const int ct = c / 4;                        // 4-channel groups (<= 256)
  const int lanes = 256 / ct;                  // row lanes
  const int tx = threadIdx.x % ct, ty = threadIdx.x / ct;
  const long long r0 = (long long)blockIdx.x * slab, r1 = min(rows, r0 + slab);
  const long long base = (long long)n * rows;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  if (ty < lanes) {
#pragma unroll 4
    for (long long r = r0 + ty; r < r1; r += lanes) {
      const long long o = (base + r) * c + tx * 4;
      float xv[4];
      gvec4<T>::ld(x + o, xv);
      if (!BWD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { s0[k] += xv[k]; s1[k] += xv[k] * xv[k]; }
      } else {
        float g[4], yv[4] = {1.f, 1.f, 1.f, 1.f};
        gvec4<T>::ld(dy + o, g);
        if (relu) gvec4<T>::ld(y + o, yv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float gk = (relu && !(yv[k] > 0.f)) ? 0.f : g[k];
          s0[k] += gk; s1[k] += gk * xv[k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][threadIdx.x][k] = s0[k]; red[1][threadIdx.x][k] = s1[k]; }
  __syncthreads();
  if (threadIdx.x < ct) {
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < lanes; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) { a[k] += red[0][q * ct + threadIdx.x][k]; b[k] += red[1][q * ct + threadIdx.x][k]; }
    float *out = partial + (((long long)n * nparts + blockIdx.x) * c + threadIdx.x * 4) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) { out[2 * k] = a[k]; out[2 * k + 1] = b[k]; }
  }
}

// ws[n][c][2] = sum over the partial rows; block = (64 channels, 16 part lanes), grid = (ceil(C/64), N)
__global__ void gn_reduce_kernel(const float *__restrict__ partial, float *__restrict__ ws, int nparts, int c) {
  __shared__ float r0[16][64], r1[16][64];
  const int n = blockIdx.y, ch = blockIdx.x * 64 + threadIdx.x;
  float a = 0.f, b = 0.f;
  if (ch < c)
    for (int p = threadIdx.y; p < nparts; p += 16) {
      const float *src = partial + (((long long)n * nparts + p) * c + ch) * 2;
      a += src[0];
      b += src[1];
    }
  r0[threadIdx.y][threadIdx.x] = a;
  r1[threadIdx.y][threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.y == 0 && ch < c) {
    float s = 0.f, q = 0.f;
    for (int k = 0; k < 16; ++k) { s += r0[k][threadIdx.x]; q += r1[k][threadIdx.x]; }
    ws[((long long)n * c + ch) * 2] = s;
    ws[((long long)n * c + ch) * 2 + 1] = q;
  }
}

__global__ void gn_finalize_fwd_kernel(const float *__restrict__ ws, float *__restrict__ mean, float *__restrict__ rstd, int ng, int c,
                                       int groups, float count, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ng) return;
  const int n = i / groups, g = i % groups, cpg = c / groups;
  float s0 = 0.f, s1 = 0.f;
  for (int k = 0; k < cpg; ++k) { s0 += ws[((long long)n * c + g * cpg + k) * 2]; s1 += ws[((long long)n * c + g * cpg + k) * 2 + 1]; }
  const float m = s0 / count;
  const float var = fmaxf(s1 / count - m * m, 0.f);
  mean[i] = m;
  rstd[i] = 1.0f / sqrtf(var + eps);
}

template <typename T>
__global__ void gn_apply_kernel(const T *__restrict__ x, T *__restrict__ y, const float *__restrict__ mean, const float *__restrict__ rstd,
                                const float *__restrict__ gamma, const float *__restrict__ beta, long long rows, int c, int groups,
                                long long total, int relu) {
  const int cpg = c / groups;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * blockDim.x * 4) {
    const int ch = (int)(i % c);
    const long long n = i / c / rows;
    float v[4];
    gvec4<T>::ld(x + i, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int sg = (int)n * groups + (ch + k) / cpg;
      v[k] = (v[k] - mean[sg]) * rstd[sg] * gamma[ch + k] + beta[ch + k];
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    gvec4<T>::st(y + i, v);
  }
}

// Round 6 fast forms of the two elementwise GroupNorm passes (the FCOS towers run them on 256 ch x 64000 rows: 8 + 8 launches per step that ran at
// 1.7 - 2.3 TB/s): 8 bf16 channels per lane in 16-byte accesses, the lane's channel group fixed by construction ((C / 8) a power of two dividing
// the block, blockIdx.y = sample) so gamma / beta / mean / rstd / coefficients leave the loop together with the 64-bit `i % c`, `i / c / rows` and the
// per-element `(ch + k) / cpg` of the general kernels.  Needs C / groups % 8 == 0 (a lane's 8 channels share one group).  Same expressions, term
// for term (this unit is compiled with -ffp-contract=off): same bits as the general kernels.
typedef __attribute__((ext_vector_type(8))) unsigned short gn_us8;
__device__ __forceinline__ void gn_ld8(const bf16s *p, float *v) {
  const gn_us8 u = *reinterpret_cast<const gn_us8 *>(p);
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = bf16_bits_to_f32(u[q]);
}
__device__ __forceinline__ void gn_st8(bf16s *p, const float *v) {
  gn_us8 u;
#pragma unroll
  for (int q = 0; q < 8; ++q) u[q] = f32_to_bf16_bits(v[q]);
  *reinterpret_cast<gn_us8 *>(p) = u;
}
static std::atomic<int> g_gn_fast{1};      // tools-only A/B switch (nerfrpn_tools.h)
extern "C" int nrpn_set_gn_fast(int on) { g_gn_fast = on ? 1 : 0; return NRPN_OK; }
static inline bool gn_fast_ok(int c, int groups, int dtype) {
  const int ct = c / 8;
  return g_gn_fast.load(std::memory_order_relaxed) && dtype == NRPN_BF16 && c % 8 == 0 && ct > 0 && ct <= 256 && (256 % ct) == 0 && (c / groups) % 8 == 0;
}

__global__ void __launch_bounds__(256) gn_apply_fast_kernel(const bf16s *__restrict__ x, bf16s *__restrict__ y, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, long long rows, int c, int groups, int relu) {
  const int ct = c / 8, n = blockIdx.y;
  const int cg = (threadIdx.x & (ct - 1)) * 8, rl = threadIdx.x / ct, lanes = 256 / ct;
  const int sg = n * groups + cg / (c / groups);
  const float m = mean[sg], rs = rstd[sg];
  float ga[8], be[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { ga[k] = gamma[cg + k]; be[k] = beta[cg + k]; }
  const long long base = (long long)n * rows;
#pragma unroll 2
  for (long long r = (long long)blockIdx.x * lanes + rl; r < rows; r += (long long)gridDim.x * lanes) {
    float v[8];
    gn_ld8(x + (base + r) * c + cg, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = (v[k] - m) * rs * ga[k] + be[k];
      if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    gn_st8(y + (base + r) * c + cg, v);
  }
}

__global__ void __launch_bounds__(256) gn_bwd_apply_fast_kernel(const bf16s *__restrict__ x, const bf16s *__restrict__ y, const bf16s *__restrict__ dy,
                                                                bf16s *__restrict__ dx, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                const float *__restrict__ gamma, const float *__restrict__ coef, long long rows, int c,
                                                                int groups, int relu) {
  const int ct = c / 8, n = blockIdx.y;
  const int cg = (threadIdx.x & (ct - 1)) * 8, rl = threadIdx.x / ct, lanes = 256 / ct;
  const int sg = n * groups + cg / (c / groups);
  const float m = mean[sg], rs = rstd[sg], cA = coef[sg * 2], cB = coef[sg * 2 + 1];
  float ga[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ga[k] = gamma[cg + k];
  const long long base = (long long)n * rows;
#pragma unroll 2
  for (long long r = (long long)blockIdx.x * lanes + rl; r < rows; r += (long long)gridDim.x * lanes) {
    const long long o = (base + r) * c + cg;
    float xv[8], gv[8], yv[8], ov[8];
    gn_ld8(x + o, xv);
    gn_ld8(dy + o, gv);
    if (relu) gn_ld8(y + o, yv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float g = (relu && !(yv[k] > 0.f)) ? 0.f : gv[k];
      const float xh = (xv[k] - m) * rs;
      ov[k] = rs * (g * ga[k] - cB - xh * cA);
    }
    gn_st8(dx + o, ov);
  }
}
static inline int gn_fast_blocks(long long rows, int c) {
  const int lanes = 256 / (c / 8);
  const long long b = (rows + lanes - 1) / lanes;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// per (sample, group): A = mean_c(g gamma xhat), B = mean_c(g gamma)
__global__ void gn_finalize_bwd_kernel(const float *__restrict__ ws, const float *__restrict__ mean, const float *__restrict__ rstd,
                                       const float *__restrict__ gamma, float *__restrict__ coef, int ng, int c, int groups, float count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ng) return;
  const int n = i / groups, g = i % groups, cpg = c / groups;
  float A = 0.f, B = 0.f;
  for (int k = 0; k < cpg; ++k) {
    const int ch = g * cpg + k;
    const float s0 = ws[((long long)n * c + ch) * 2], s1 = ws[((long long)n * c + ch) * 2 + 1];
    A += gamma[ch] * (s1 - mean[i] * s0) * rstd[i];
    B += gamma[ch] * s0;
  }
  coef[i * 2] = A / count;
  coef[i * 2 + 1] = B / count;
}

__global__ void gn_param_grad_kernel(const float *__restrict__ ws, const float *__restrict__ mean, const float *__restrict__ rstd,
                                     float *__restrict__ dgamma, float *__restrict__ dbeta, int n, int c, int groups, int acc) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= c) return;
  const int g = ch / (c / groups);
  float dg = 0.f, db = 0.f;
  for (int s = 0; s < n; ++s) {
    const float s0 = ws[((long long)s * c + ch) * 2], s1 = ws[((long long)s * c + ch) * 2 + 1];
    dg += (s1 - mean[s * groups + g] * s0) * rstd[s * groups + g];
    db += s0;
  }
  dgamma[ch] = acc ? dgamma[ch] + dg : dg;       // acc: added to the gradient-arena slots instead of a torch add per parameter
  dbeta[ch] = acc ? dbeta[ch] + db : db;
}

template <typename T>
__global__ void gn_bwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy, T *__restrict__ dx,
                                    const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                    const float *__restrict__ coef, long long rows, int c, int groups, long long total, int relu) {
  const int cpg = c / groups;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * blockDim.x * 4) {
    const int ch = (int)(i % c);
    const long long n = i / c / rows;
    float xv[4], gv[4], yv[4] = {1.f, 1.f, 1.f, 1.f}, o[4];
    gvec4<T>::ld(x + i, xv);
    gvec4<T>::ld(dy + i, gv);
    if (relu) gvec4<T>::ld(y + i, yv);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int sg = (int)n * groups + (ch + k) / cpg;
      const float g = (relu && !(yv[k] > 0.f)) ? 0.f : gv[k];
      const float xh = (xv[k] - mean[sg]) * rstd[sg];
      o[k] = rstd[sg] * (g * gamma[ch + k] - coef[sg * 2 + 1] - xh * coef[sg * 2]);
    }
    gvec4<T>::st(dx + i, o);
  }
}

// workspace: sums [n][c][2] | group coefficients [n][groups][2] | partial rows [n][kGnMaxParts][c][2]
extern "C" size_t nrpn_groupnorm_workspace_bytes(int n, int c, int groups) {
  return ((size_t)n * c * 2 + (size_t)n * groups * 2 + (size_t)n * kGnMaxParts * c * 2) * 4;
}

extern "C" int nrpn_groupnorm_fwd(const void *x, void *y, const float *gamma, const float *beta, float *mean, float *rstd, int n,
                                  int64_t rows, int c, int groups, float eps, int relu, int dtype, void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(x && y && gamma && beta && mean && rstd && workspace && n > 0 && rows > 0 && c > 0 && groups > 0 && c % groups == 0,
               "groupnorm_fwd: bad args");
  hipStream_t st = as_stream(stream);
  NRPN_REQUIRE(c % 4 == 0 && c <= 1024, "groupnorm: C=%d must be a multiple of 4 and <= 1024", c);
  float *ws = reinterpret_cast<float *>(workspace);
  float *partial = ws + (size_t)n * c * 2 + (size_t)n * groups * 2;
  const int slab = gn_slab(rows), nparts = (int)cdiv64(rows, slab);
  DISPATCH_T(dtype, hipLaunchKernelGGL((gn_partial_kernel<T, false>), dim3(nparts, n), dim3(256), 0, st, (const T *)x, (const T *)nullptr,
                                       (const T *)nullptr, partial, (long long)rows, c, 0, nparts, slab));
  hipLaunchKernelGGL(gn_reduce_kernel, dim3((c + 63) / 64, n), dim3(64, 16), 0, st, (const float *)partial, ws, nparts, c);
  const int ng = n * groups;
  hipLaunchKernelGGL(gn_finalize_fwd_kernel, dim3((ng + 63) / 64), dim3(64), 0, st, ws, mean, rstd, ng, c, groups,
                     (float)((double)rows * (c / groups)), eps);
  const long long total = (long long)n * rows * c;
  if (gn_fast_ok(c, groups, dtype)) {
    hipLaunchKernelGGL(gn_apply_fast_kernel, dim3(gn_fast_blocks(rows, c), n), dim3(256), 0, st, (const bf16s *)x, (bf16s *)y, mean, rstd, gamma, beta,
                       (long long)rows, c, groups, relu);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(ew_blocks(total / 4)), dim3(256), 0, st, (const T *)x, (T *)y, mean, rstd, gamma,
                                         beta, (long long)rows, c, groups, total, relu));
  }
  NRPN_LAUNCH_CHECK("groupnorm_fwd");
  return NRPN_OK;
}

extern "C" int nrpn_groupnorm_bwd(const void *x, const void *y, const void *dy, void *dx, const float *gamma, const float *mean,
                                  const float *rstd, float *dgamma, float *dbeta, int n, int64_t rows, int c, int groups, int relu, int dtype,
                                  int accumulate_params, void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(x && dy && dx && gamma && mean && rstd && dgamma && dbeta && workspace && (!relu || y) && n > 0 && rows > 0 && c > 0 &&
                   groups > 0 && c % groups == 0, "groupnorm_bwd: bad args");
  hipStream_t st = as_stream(stream);
  NRPN_REQUIRE(c % 4 == 0 && c <= 1024, "groupnorm: C=%d must be a multiple of 4 and <= 1024", c);
  float *ws = reinterpret_cast<float *>(workspace);
  float *coef = ws + (size_t)n * c * 2;
  float *partial = coef + (size_t)n * groups * 2;
  const int slab = gn_slab(rows), nparts = (int)cdiv64(rows, slab);
  DISPATCH_T(dtype, hipLaunchKernelGGL((gn_partial_kernel<T, true>), dim3(nparts, n), dim3(256), 0, st, (const T *)x, (const T *)y, (const T *)dy,
                                       partial, (long long)rows, c, relu, nparts, slab));
  hipLaunchKernelGGL(gn_reduce_kernel, dim3((c + 63) / 64, n), dim3(64, 16), 0, st, (const float *)partial, ws, nparts, c);
  const int ng = n * groups;
  hipLaunchKernelGGL(gn_finalize_bwd_kernel, dim3((ng + 63) / 64), dim3(64), 0, st, ws, mean, rstd, gamma, coef, ng, c, groups,
                     (float)((double)rows * (c / groups)));
  hipLaunchKernelGGL(gn_param_grad_kernel, dim3((c + 255) / 256), dim3(256), 0, st, ws, mean, rstd, dgamma, dbeta, n, c, groups,
                     accumulate_params ? 1 : 0);
  const long long total = (long long)n * rows * c;
  if (gn_fast_ok(c, groups, dtype)) {
    hipLaunchKernelGGL(gn_bwd_apply_fast_kernel, dim3(gn_fast_blocks(rows, c), n), dim3(256), 0, st, (const bf16s *)x, (const bf16s *)y, (const bf16s *)dy,
                       (bf16s *)dx, mean, rstd, gamma, coef, (long long)rows, c, groups, relu);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3(ew_blocks(total / 4)), dim3(256), 0, st, (const T *)x, (const T *)y,
                                         (const T *)dy, (T *)dx, mean, rstd, gamma, coef, (long long)rows, c, groups, total, relu));
  }
  NRPN_LAUNCH_CHECK("groupnorm_bwd");
  return NRPN_OK;
}

// =====================================================================================================================
// head epilogue
// =====================================================================================================================
__global__ void fcos_head_out_kernel(const float *__restrict__ cls_out, const float *__restrict__ box_out, int wrows,
                                     const float *__restrict__ scale, float stride_mul, int norm_reg, int D, int ctr_on_reg, long long rows,
                                     float *__restrict__ logits, float *__restrict__ reg, float *__restrict__ ctr) {
  const float sc = scale[0];
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
    logits[r] = cls_out[r * wrows];
    ctr[r] = ctr_on_reg ? box_out[r * wrows + D] : cls_out[r * wrows + 1];
    for (int j = 0; j < D; ++j) {
      float v = box_out[r * wrows + j] * sc;
      if (norm_reg) { if (j < 6) v = fmaxf(v, 0.f) * stride_mul; }
      else v = expf(v);
      reg[r * D + j] = v;
    }
  }
}

// Deterministic grid-wide sum of one value per workgroup: buf = [0] result, [1] ticket (zero before the launch; left zero), [2 + block]
// partials.  Every workgroup stores its partial, the one drawing the last ticket adds them in block order (cf. sumsq_kernel): the result
// does not depend on the order the workgroups finish in.  All threads of the workgroup must call it; `mine` is taken from thread 0.
constexpr int kOrderedSumFloats = 2 + 1024;
__device__ __forceinline__ void ordered_block_sum(float *buf, float mine) {
  __shared__ bool last;
  __shared__ float part[256];
  if (threadIdx.x == 0) {
    buf[2 + blockIdx.x] = mine;
    __threadfence();
    last = atomicAdd(reinterpret_cast<unsigned *>(buf + 1), 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (int k = threadIdx.x; k < (int)gridDim.x; k += blockDim.x) t += buf[2 + k];      // fixed assignment of partials to threads ...
  part[threadIdx.x] = t;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s]; __syncthreads(); }      // ... fixed tree
  if (threadIdx.x == 0) { buf[0] = part[0]; reinterpret_cast<unsigned *>(buf)[1] = 0u; }
}

__global__ void __launch_bounds__(256) fcos_head_out_bwd_kernel(const float *__restrict__ box_out, int wrows, const float *__restrict__ scale,
                                                                float stride_mul, int norm_reg, int D, int ctr_on_reg, long long rows,
                                                                const float *__restrict__ d_logits, const float *__restrict__ d_reg,
                                                                const float *__restrict__ d_ctr, float *__restrict__ d_cls_out,
                                                                float *__restrict__ d_box_out, float *__restrict__ d_scale) {
  __shared__ float red[256];
  const float sc = scale[0];
  float ds = 0.f;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
    float *dc = d_cls_out + r * wrows, *db = d_box_out + r * wrows;
    for (int j = 0; j < wrows; ++j) { dc[j] = 0.f; db[j] = 0.f; }
    dc[0] = d_logits ? d_logits[r] : 0.f;
    const float dct = d_ctr ? d_ctr[r] : 0.f;
    if (ctr_on_reg) db[D] = dct; else dc[1] = dct;
    if (d_reg) {
      for (int j = 0; j < D; ++j) {
        const float raw = box_out[r * wrows + j];
        float dv = d_reg[r * D + j];
        if (norm_reg) { if (j < 6) dv = (raw * sc > 0.f) ? dv * stride_mul : 0.f; }
        else dv *= expf(raw * sc);
        db[j] = dv * sc;
        ds += dv * raw;
      }
    }
  }
  red[threadIdx.x] = ds;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  ordered_block_sum(d_scale, red[0]);
}

extern "C" int nrpn_fcos_head_out_f32(const float *cls_out, const float *box_out, int wrows, const float *scale, float stride_mul,
                                      int norm_reg, int reg_dim, int ctr_on_reg, int64_t rows, float *logits, float *reg, float *ctr,
                                      nrpn_stream_t stream) {
  NRPN_REQUIRE(cls_out && box_out && scale && logits && reg && ctr && rows > 0 && (reg_dim == 6 || reg_dim == 8) && wrows > reg_dim,
               "fcos_head_out: bad args");
  hipLaunchKernelGGL(fcos_head_out_kernel, dim3(ew_blocks(rows)), dim3(256), 0, as_stream(stream), cls_out, box_out, wrows, scale, stride_mul,
                     norm_reg, reg_dim, ctr_on_reg, (long long)rows, logits, reg, ctr);
  NRPN_LAUNCH_CHECK("fcos_head_out");
  return NRPN_OK;
}

extern "C" int nrpn_fcos_reduce_floats(void) { return kOrderedSumFloats; }

extern "C" int nrpn_fcos_head_out_bwd_f32(const float *box_out, int wrows, const float *scale, float stride_mul, int norm_reg, int reg_dim,
                                          int ctr_on_reg, int64_t rows, const float *d_logits, const float *d_reg, const float *d_ctr,
                                          float *d_cls_out, float *d_box_out, float *d_scale, nrpn_stream_t stream) {
  NRPN_REQUIRE(box_out && scale && d_cls_out && d_box_out && d_scale && rows > 0 && (reg_dim == 6 || reg_dim == 8) && wrows > reg_dim,
               "fcos_head_out_bwd: bad args");
  hipLaunchKernelGGL(fcos_head_out_bwd_kernel, dim3(ew_blocks(rows, 1024)), dim3(256), 0, as_stream(stream), box_out, wrows, scale, stride_mul,
                     norm_reg, reg_dim, ctr_on_reg, (long long)rows, d_logits, d_reg, d_ctr, d_cls_out, d_box_out, d_scale);
  NRPN_LAUNCH_CHECK("fcos_head_out_bwd");
  return NRPN_OK;
}

// =====================================================================================================================
// geometry of the flattened location list: level-major, then scene, then voxel (x, y, z) with z fastest
// =====================================================================================================================
constexpr int kMaxLevels = 8, kMaxScenes = 64;

struct FcosGeom {
  int levels, n;
  int dims[kMaxLevels][3];
  int stride[kMaxLevels];
  long long off[kMaxLevels + 1];      // start of each level in the flat list
  float ori[kMaxScenes][3];           // un-padded scene sizes (padding mask), or +inf
  int gt_off[kMaxScenes + 1];
};

static int fill_fcos_geom(FcosGeom &g, int n, int levels, const int32_t *dims, const int32_t *strides, const float *ori_sizes,
                          const int32_t *gt_offsets) {
  if (n <= 0 || n > kMaxScenes || levels <= 0 || levels > kMaxLevels || !dims || !strides)
    return nrpn_fail(NRPN_ERR_ARG, "fcos: need 1..%d scenes and 1..%d levels", kMaxScenes, kMaxLevels);
  g.levels = levels; g.n = n;
  g.off[0] = 0;
  for (int l = 0; l < levels; ++l) {
    for (int d = 0; d < 3; ++d) g.dims[l][d] = dims[l * 3 + d];
    g.stride[l] = strides[l];
    g.off[l + 1] = g.off[l] + (long long)n * dims[l * 3] * dims[l * 3 + 1] * dims[l * 3 + 2];
  }
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) g.ori[i][d] = ori_sizes ? ori_sizes[i * 3 + d] : INFINITY;
  for (int i = 0; i <= n; ++i) g.gt_off[i] = gt_offsets ? gt_offsets[i] : 0;
  return 0;
}

// flat index -> level, scene, location (compute_locations_per_level, fcos.py:233-250)
__device__ __forceinline__ void fcos_locate(const FcosGeom &g, long long i, int &level, int &scene, float &x, float &y, float &z) {
  level = 0;
  while (level + 1 < g.levels && i >= g.off[level + 1]) ++level;
  long long v = i - g.off[level];
  const int dz = g.dims[level][2], dy = g.dims[level][1], dx = g.dims[level][0];
  const int iz = (int)(v % dz); v /= dz;
  const int iy = (int)(v % dy); v /= dy;
  const int ix = (int)(v % dx);
  scene = (int)(v / dx);
  const int s = g.stride[level];
  x = (float)(ix * s) + (float)(s / 2);
  y = (float)(iy * s) + (float)(s / 2);
  z = (float)(iz * s) + (float)(s / 2);
}

// =====================================================================================================================
// training targets
// =====================================================================================================================
// per-GT quantities of encode_fcos_obb that do not depend on the location (utils.py:65-105): footprint AABB, alpha, beta
__global__ void fcos_gt_summary_kernel(const float *__restrict__ gt, int count, int width, float *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float *o = out + i * 8;
  if (width == 6) {
    for (int j = 0; j < 6; ++j) o[j] = gt[i * 6 + j];
    o[6] = 0.f; o[7] = 0.f;
    return;
  }
  const float *b = gt + i * 7;
  float X[4], Y[4];
  geo::corners2d(b[0], b[1], b[3], b[4], b[6], X, Y);
  float xmax = X[0], xmin = X[0], ymax = Y[0], ymin = Y[0];
  for (int k = 1; k < 4; ++k) { xmax = fmaxf(xmax, X[k]); xmin = fminf(xmin, X[k]); ymax = fmaxf(ymax, Y[k]); ymin = fminf(ymin, Y[k]); }
  float vx = -INFINITY, vy = INFINITY;
  for (int k = 0; k < 4; ++k) {
    const float xt = (ymax - Y[k] > 0.1f) ? -1e6f : X[k];
    const float yt = (xmax - X[k] > 0.1f) ? 1e6f : Y[k];
    vx = fmaxf(vx, xt);
    vy = fminf(vy, yt);
  }
  // torch.isclose(a, b): |a - b| <= 1e-8 + 1e-5 * |b|
  const bool close = (fabsf(vx - xmax) <= 1e-8f + 1e-5f * fabsf(xmax)) && (fabsf(vy - ymin) <= 1e-8f + 1e-5f * fabsf(ymin));
  if (close) { vx = xmax; vy = ymin; }
  o[0] = xmin; o[1] = ymin; o[2] = b[2] - b[5] / 2.f; o[3] = xmax; o[4] = ymax; o[5] = b[2] + b[5] / 2.f;
  o[6] = (vx - b[0]) / (xmax - xmin);
  o[7] = (vy - b[1]) / (ymax - ymin);
}

extern "C" int nrpn_fcos_gt_summary_f32(const float *gt, int count, int width, float *summary, nrpn_stream_t stream) {
  NRPN_REQUIRE(gt && summary && count > 0 && (width == 6 || width == 7), "fcos_gt_summary: bad args");
  hipLaunchKernelGGL(fcos_gt_summary_kernel, dim3((count + 63) / 64), dim3(64), 0, as_stream(stream), gt, count, width, summary);
  NRPN_LAUNCH_CHECK("fcos_gt_summary");
  return NRPN_OK;
}

__global__ void __launch_bounds__(256) fcos_targets_kernel(const float *__restrict__ summary, FcosGeom g, float radius, int norm_reg, int D,
                                                           signed char *__restrict__ labels, float *__restrict__ reg_targets,
                                                           int *__restrict__ num_pos) {
  const float kInf = 100000000.0f;
  const float soi[4][2] = {{-1.f, 16.f}, {16.f, 32.f}, {32.f, 64.f}, {64.f, kInf}};     // loss.py:272-277
  const long long total = g.off[g.levels];
  int local_pos = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int level, scene;
    float x, y, z;
    fcos_locate(g, i, level, scene, x, y, z);
    float *rt = reg_targets + i * D;
    if (!(x < g.ori[scene][0] && y < g.ori[scene][1] && z < g.ori[scene][2])) {      // padding of a batched scene: excluded
      labels[i] = -1;
      for (int j = 0; j < D; ++j) rt[j] = 0.f;
      continue;
    }
    const int g0 = g.gt_off[scene], g1 = g.gt_off[scene + 1];
    const float s = (float)g.stride[level];
    const float rad = s * radius;
    const float lo = soi[level < 4 ? level : 3][0], hi = soi[level < 4 ? level : 3][1];
    float best = kInf;
    int which = g0;
    for (int k = g0; k < g1; ++k) {
      const float *a = summary + k * 8;
      const float l = x - a[0], t = y - a[1], f = z - a[2], r = a[3] - x, b = a[4] - y, ba = a[5] - z;
      bool inside;
      if (radius > 0.f) {          // get_sample_region, loss.py:209-268
        const float cx = (a[0] + a[3]) / 2.f, cy = (a[1] + a[4]) / 2.f, cz = (a[2] + a[5]) / 2.f;
        const float x0 = (cx - rad > a[0]) ? cx - rad : a[0], y0 = (cy - rad > a[1]) ? cy - rad : a[1], z0 = (cz - rad > a[2]) ? cz - rad : a[2];
        const float x1 = (cx + rad > a[3]) ? a[3] : cx + rad, y1 = (cy + rad > a[4]) ? a[4] : cy + rad, z1 = (cz + rad > a[5]) ? a[5] : cz + rad;
        inside = fminf(fminf(fminf(x - x0, y - y0), fminf(x1 - x, y1 - y)), fminf(z - z0, z1 - z)) > 0.f;
      } else {
        inside = fminf(fminf(fminf(l, t), fminf(f, r)), fminf(b, ba)) > 0.f;
      }
      const float mx = fmaxf(fmaxf(fmaxf(l, t), fmaxf(f, r)), fmaxf(b, ba));
      const bool cared = mx >= lo && mx <= hi;
      float vol = (a[3] - a[0]) * (a[4] - a[1]) * (a[5] - a[2]);
      if (!inside || !cared) vol = kInf;
      if (vol < best) { best = vol; which = k; }
    }
    const bool pos = g1 > g0 && best != kInf;
    labels[i] = pos ? 1 : 0;
    local_pos += pos ? 1 : 0;
    if (g1 > g0) {
      const float *a = summary + which * 8;
      const float dn = norm_reg ? s : 1.0f;
      rt[0] = (x - a[0]) / dn; rt[1] = (y - a[1]) / dn; rt[2] = (z - a[2]) / dn;
      rt[3] = (a[3] - x) / dn; rt[4] = (a[4] - y) / dn; rt[5] = (a[5] - z) / dn;
      if (D == 8) { rt[6] = a[6]; rt[7] = a[7]; }
    } else {
      for (int j = 0; j < D; ++j) rt[j] = 0.f;
    }
  }
  if (local_pos) atomicAdd(num_pos, local_pos);
}

extern "C" int nrpn_fcos_targets_f32(const float *summary, const int32_t *gt_offsets, int n, int levels, const int32_t *dims,
                                     const int32_t *strides, const float *ori_sizes, float radius, int norm_reg, int reg_dim, int8_t *labels,
                                     float *reg_targets, int32_t *num_pos, nrpn_stream_t stream) {
  FcosGeom g;
  if (int rc = fill_fcos_geom(g, n, levels, dims, strides, ori_sizes, gt_offsets)) return rc;
  NRPN_REQUIRE(gt_offsets && labels && reg_targets && num_pos && (reg_dim == 6 || reg_dim == 8) && (summary || gt_offsets[n] == 0),
               "fcos_targets: bad args");
  hipStream_t st = as_stream(stream);
  NRPN_HIP(hipMemsetAsync(num_pos, 0, 4, st));
  hipLaunchKernelGGL(fcos_targets_kernel, dim3(ew_blocks(g.off[levels], 2048)), dim3(256), 0, st, summary, g, radius, norm_reg, reg_dim,
                     reinterpret_cast<signed char *>(labels), reg_targets, num_pos);
  NRPN_LAUNCH_CHECK("fcos_targets");
  return NRPN_OK;
}

// =====================================================================================================================
// sigmoid focal loss (alpha, gamma = 2), sum reduction, with d loss / d logit (torchvision.ops.sigmoid_focal_loss)
// =====================================================================================================================
__device__ __forceinline__ float softplusf(float v) { return fmaxf(v, 0.f) + log1pf(expf(-fabsf(v))); }

__global__ void __launch_bounds__(256) fcos_focal_kernel(const float *__restrict__ logits, const signed char *__restrict__ labels,
                                                         long long count, float alpha, float *__restrict__ loss_sum,
                                                         float *__restrict__ dlogits) {
  __shared__ float red[256];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
    const int t = labels[i];
    float d = 0.f;
    if (t >= 0) {
      const float xv = logits[i];
      const float p = 1.0f / (1.0f + expf(-xv));
      if (t > 0) {
        const float logp = -softplusf(-xv), q = 1.0f - p;
        acc += -alpha * q * q * logp;
        d = alpha * q * q * (2.0f * p * logp - q);
      } else {
        const float log1mp = -softplusf(xv), q = 1.0f - p;
        acc += -(1.0f - alpha) * p * p * log1mp;
        d = (1.0f - alpha) * p * p * (p - 2.0f * q * log1mp);
      }
    }
    if (dlogits) dlogits[i] = d;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  ordered_block_sum(loss_sum, red[0]);
}

extern "C" int nrpn_fcos_focal_f32(const float *logits, const int8_t *labels, int64_t count, float alpha, float *loss_sum, float *dlogits,
                                   nrpn_stream_t stream) {
  NRPN_REQUIRE(logits && labels && loss_sum && count > 0, "fcos_focal: bad args");
  hipStream_t st = as_stream(stream);
  NRPN_HIP(hipMemsetAsync(loss_sum, 0, 8, st));      // result + ticket
  hipLaunchKernelGGL(fcos_focal_kernel, dim3(ew_blocks(count, 1024)), dim3(256), 0, st, logits, reinterpret_cast<const signed char *>(labels),
                     (long long)count, alpha, loss_sum, dlogits);
  NRPN_LAUNCH_CHECK("fcos_focal");
  return NRPN_OK;
}

// =====================================================================================================================
// inference: scores and candidate decode
// =====================================================================================================================
// score = sigmoid(cls) * sigmoid(ctr) for candidates (un-padded location, sigmoid(cls) > pre_nms_thresh), else -1
__global__ void fcos_scores_kernel(const float *__restrict__ logits, const float *__restrict__ ctr, FcosGeom g, float thresh,
                                   float *__restrict__ scores) {
  const long long total = g.off[g.levels];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int level, scene;
    float x, y, z;
    fcos_locate(g, i, level, scene, x, y, z);
    const float c = 1.0f / (1.0f + expf(-logits[i]));
    const bool cand = (x < g.ori[scene][0] && y < g.ori[scene][1] && z < g.ori[scene][2]) && c > thresh;
    scores[i] = cand ? c * (1.0f / (1.0f + expf(-ctr[i]))) : -1.0f;
  }
}

extern "C" int nrpn_fcos_scores_f32(const float *logits, const float *ctr, int n, int levels, const int32_t *dims, const int32_t *strides,
                                    const float *ori_sizes, float pre_nms_thresh, float *scores, nrpn_stream_t stream) {
  FcosGeom g;
  if (int rc = fill_fcos_geom(g, n, levels, dims, strides, ori_sizes, nullptr)) return rc;
  NRPN_REQUIRE(logits && ctr && scores, "fcos_scores: null pointer");
  hipLaunchKernelGGL(fcos_scores_kernel, dim3(ew_blocks(g.off[levels])), dim3(256), 0, as_stream(stream), logits, ctr, g, pre_nms_thresh, scores);
  NRPN_LAUNCH_CHECK("fcos_scores");
  return NRPN_OK;
}

// decode_fcos_obb (utils.py:12-62)
__device__ __forceinline__ void fcos_decode_obb(float lx, float ly, float lz, const float *r, float *o) {
  const float x0 = lx - r[0], y0 = ly - r[1], z0 = lz - r[2], x1 = lx + r[3], y1 = ly + r[4], z1 = lz + r[5];
  float vx = (x1 + x0) / 2.f + r[6] * (x1 - x0);
  float vy = (y1 + y0) / 2.f + r[7] * (y1 - y0);
  vx = fminf(fmaxf(vx, x0), x1);
  vy = fminf(fmaxf(vy, y0), y1);
  const float cx = (x0 + x1) / 2.f, cy = (y0 + y1) / 2.f, cz = (z0 + z1) / 2.f;
  float ax = vx - cx, ay = y1 - cy, bx = x1 - cx, by = vy - cy;
  const float d0 = sqrtf(ax * ax + ay * ay), d1 = sqrtf(bx * bx + by * by);
  const float dm = fmaxf(d0, d1);
  ax = ax / (d0 + 1e-7f) * dm + cx; ay = ay / (d0 + 1e-7f) * dm + cy;
  bx = bx / (d1 + 1e-7f) * dm + cx; by = by / (d1 + 1e-7f) * dm + cy;
  const float ex = ax - bx, ey = ay - by;
  const float len = sqrtf(ex * ex + ey * ey);
  float mx = (ax + bx) / 2.f - cx, my = (ay + by) / 2.f - cy;
  const float wid = sqrtf(mx * mx + my * my) * 2.f;
  if (mx == 0.f && my == 0.f) mx = 1e-7f;
  o[0] = cx; o[1] = cy; o[2] = cz; o[3] = wid; o[4] = len; o[5] = z1 - z0; o[6] = atan2f(my, mx);
}

// idx: flat location index of each candidate (or < 0); score: cls*ctr (or < 0).  Writes the box, sqrt(score) (or -1 when the
// candidate is missing / smaller than min_size) and the pyramid level.
__global__ void fcos_decode_kernel(const int *__restrict__ idx, const float *__restrict__ score, long long count, long long seg_len,
                                   const float *__restrict__ reg, FcosGeom g, int D, float min_size, float *__restrict__ boxes,
                                   float *__restrict__ out_scores, float *__restrict__ out_levels) {
  const int W = D == 8 ? 7 : 6;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
    float *o = boxes + i * W;
    const float sc = score[i];
    const int seg = (int)(i / seg_len);                 // segments are (level, scene) pairs, level-major
    const int level = seg / g.n, scene = seg % g.n;
    out_levels[i] = (float)level;
    if (idx[i] < 0 || !(sc >= 0.f)) {
      for (int j = 0; j < W; ++j) o[j] = 0.f;
      out_scores[i] = -1.f;
      continue;
    }
    const long long flat = g.off[level] + (long long)scene * g.dims[level][0] * g.dims[level][1] * g.dims[level][2] + idx[i];
    int lv, sn;
    float x, y, z;
    fcos_locate(g, flat, lv, sn, x, y, z);
    const float *r = reg + flat * D;
    bool keep;
    if (D == 6) {
      o[0] = fminf(fmaxf(x - r[0], 0.f), g.ori[scene][0]); o[1] = fminf(fmaxf(y - r[1], 0.f), g.ori[scene][1]);
      o[2] = fminf(fmaxf(z - r[2], 0.f), g.ori[scene][2]); o[3] = fminf(fmaxf(x + r[3], 0.f), g.ori[scene][0]);
      o[4] = fminf(fmaxf(y + r[4], 0.f), g.ori[scene][1]); o[5] = fminf(fmaxf(z + r[5], 0.f), g.ori[scene][2]);
      keep = (o[3] - o[0] >= min_size) && (o[4] - o[1] >= min_size) && (o[5] - o[2] >= min_size);
    } else {
      fcos_decode_obb(x, y, z, r, o);
      keep = o[3] >= min_size && o[4] >= min_size && o[5] >= min_size;
    }
    out_scores[i] = keep ? sqrtf(sc) : -1.f;
  }
}

extern "C" int nrpn_fcos_decode_f32(const int32_t *idx, const float *score, int64_t count, int64_t seg_len, const float *reg, int n,
                                    int levels, const int32_t *dims, const int32_t *strides, const float *ori_sizes, int reg_dim,
                                    float min_size, float *boxes, float *out_scores, float *out_levels, nrpn_stream_t stream) {
  FcosGeom g;
  if (int rc = fill_fcos_geom(g, n, levels, dims, strides, ori_sizes, nullptr)) return rc;
  NRPN_REQUIRE(idx && score && reg && boxes && out_scores && out_levels && count > 0 && seg_len > 0 && count == seg_len * n * levels &&
                   ori_sizes && (reg_dim == 6 || reg_dim == 8), "fcos_decode: bad args (count must be levels * scenes * seg_len)");
  hipLaunchKernelGGL(fcos_decode_kernel, dim3(ew_blocks(count)), dim3(256), 0, as_stream(stream), idx, score, (long long)count,
                     (long long)seg_len, reg, g, reg_dim, min_size, boxes, out_scores, out_levels);
  NRPN_LAUNCH_CHECK("fcos_decode");
  return NRPN_OK;
}

// HBM-bound companions of the conv kernels (gfx950): BatchNorm3d statistics / apply / backward, ReLU backward,
// MaxPool3d, nearest-upsample-add (FPN top-down), layout + dtype conversion, and the flat-arena AdamW step.
// All tensors channels-last [rows = N*X*Y*Z][C]; each lane handles 4 consecutive channels (16-byte fp32 / 8-byte bf16
// accesses, fully coalesced along C); per-channel reductions are block partials in fp32 finished in fp64.
//
// Replaces BatchNorm3d / ReLU / MaxPool3d / F.interpolate+add in reference feature_extractor.py:337-358, fpn.py:150-155
// and clip_grad_norm_ + AdamW of run_rpn.py:345-349,390-395.
#include "common.h"
#include <atomic>

typedef unsigned short bf16s;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned short us4;

// bf16 stores of this file round on the conversion unit (fptrunc -> v_cvt_pk_bf16_f32, two values per instruction) instead of common.h's
// integer routine (~7 VALU instructions per value): the same round-to-nearest-even result for every non-NaN input (these kernels keep
// fp32 denormals), and the companions are VALU-, not bandwidth-limited when they share the chip with the weight-gradient stream.
template <typename T> struct vec4;
template <> struct vec4<float> {
  static __device__ __forceinline__ f4 ld(const float *p) { return *reinterpret_cast<const f4 *>(p); }
  static __device__ __forceinline__ void st(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
};
template <> struct vec4<bf16s> {
  static __device__ __forceinline__ f4 ld(const bf16s *p) {
    const us4 u = *reinterpret_cast<const us4 *>(p);
    f4 v = {bf16_bits_to_f32(u[0]), bf16_bits_to_f32(u[1]), bf16_bits_to_f32(u[2]), bf16_bits_to_f32(u[3])};
    return v;
  }
  static __device__ __forceinline__ void st(bf16s *p, f4 v) {
    typedef __attribute__((ext_vector_type(4))) __bf16 b4;
    *reinterpret_cast<us4 *>(p) = __builtin_bit_cast(us4, __builtin_convertvector(v, b4));      // v_cvt_pk_bf16_f32, see below
  }
};

// V consecutive channels per thread: 4 (8 / 16-byte accesses) or, for bf16 with C % 8 == 0, 8 (16-byte accesses): half the
// instructions per byte on kernels that are issue-bound (27 window positions per output of the 3/2/1 pool)
template <typename T, int V> struct vecv;
template <typename T> struct vecv<T, 4> {
  static __device__ __forceinline__ void ld(const T *p, float *o) { const f4 v = vec4<T>::ld(p); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
  static __device__ __forceinline__ void st(T *p, const float *o) { vec4<T>::st(p, f4{o[0], o[1], o[2], o[3]}); }
};
template <> struct vecv<bf16s, 8> {
  typedef __attribute__((ext_vector_type(8))) unsigned short us8;
  static __device__ __forceinline__ void ld(const bf16s *p, float *o) {
    const us8 u = *reinterpret_cast<const us8 *>(p);
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = bf16_bits_to_f32(u[q]);
  }
  static __device__ __forceinline__ void st(bf16s *p, const float *o) {
    typedef __attribute__((ext_vector_type(8))) float f8;
    typedef __attribute__((ext_vector_type(8))) __bf16 b8;
    const f8 v = {o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
    *reinterpret_cast<us8 *>(p) = __builtin_bit_cast(us8, __builtin_convertvector(v, b8));
  }
};

#define DISPATCH_T(dtype, ...)                       \
  if ((dtype) == NRPN_F32) { typedef float T; __VA_ARGS__; } \
  else { typedef bf16s T; __VA_ARGS__; }

// =====================================================================================================================
// per-channel reductions: slabs of 256 rows per block; thread = (4-channel group, row lane)
// =====================================================================================================================
// rows per block of the channel reductions: ~1024 blocks for any tensor (fixed 256-row slabs left 3/4 of the chip idle on the
// 64000-row maps), at least 64 rows so the finalize kernels read a bounded number of partials.  Round 6 sweep of the block target on one
// MI355X (bn_stats / bn_backward on 256 ch x 64000 rows, us incl. the finish): 512 blocks 11.6 / 41.2, 1024 12.7 / 38.4, 2048 17.6 / 47.8,
// 4096 25.7 / 60.2 -- more partial rows cost the latency-bound finish more than the reduction gains
static inline int slab_rows(long long rows) { const long long s = (rows + 1023) / 1024; return (int)(s < 64 ? 64 : s); }

// MODE 0: (sum x, sum x^2)           MODE 1: (sum dy', sum dy' * xhat) with dy' = dy * (y > 0 if relu)
// V channels per lane: 4, or 8 for bf16 with C % 8 == 0 (16-byte loads: half the load instructions and address arithmetic per byte; round 5 --
// the 8-byte form ran at 0.27-0.39 of the HBM rate on the 64000-row maps).
// rows in flight per lane.  Round 6 sweep (-DNRPN_CHAN_UNROLL, one MI355X; bn_backward on 64 ch x 512000 / 256 ch x 64000 rows, us; bench step, ms):
// 2: 61.6 / 39.1, 8.70-8.75;  4: 60.4 / 39.1, 8.72-8.78;  8: 70.0 / 43.9, 8.82-8.86;  16: 97.5 / 59.4, 8.94 -- more loads in flight cost occupancy
#ifndef NRPN_CHAN_UNROLL
#define NRPN_CHAN_UNROLL 4
#endif
template <typename T, int MODE, int V>
__global__ void __launch_bounds__(256)
chan_partial_kernel(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy, long long rows, int c,
                    const float *__restrict__ mean, const float *__restrict__ var, float eps, int relu, float *__restrict__ partial,
                    int kSlab, const float *__restrict__ gamma = nullptr, const float *__restrict__ beta = nullptr) {
  __shared__ float red[2][256][V];
  // channels are tiled over blockIdx.y in chunks of `cw` (<= 1024) so any C that is a multiple of V works
  const int cw = min(c, 1024), coff = blockIdx.y * 1024;
  x += coff; y = y ? y + coff : y; dy = dy ? dy + coff : dy;
  if (MODE == 1) { mean += coff; var += coff; if (beta) { gamma += coff; beta += coff; } }
  const int ct = min(cw, c - coff) / V;       // channel groups of this tile (<= 256)
  const int ty_n = 256 / ct;                  // row lanes
  const int tx = threadIdx.x % ct, ty = threadIdx.x / ct;
  const long long r0 = (long long)blockIdx.x * kSlab, r1 = min(rows, r0 + kSlab);
  float s0[V], s1[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { s0[k] = 0.f; s1[k] = 0.f; }
  if (ty < ty_n) {
    float mu[V], is[V], ga[V], be[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { mu[k] = 0.f; is[k] = 1.f; ga[k] = 1.f; be[k] = 0.f; }
    if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < V; ++k) {
        mu[k] = mean[tx * V + k];
        is[k] = 1.0f / sqrtf(var[tx * V + k] + eps);
        if (beta) { ga[k] = gamma[tx * V + k]; be[k] = beta[tx * V + k]; }
      }
    }
#pragma unroll NRPN_CHAN_UNROLL
    for (long long r = r0 + ty; r < r1; r += ty_n) {
      float xv[V];
      vecv<T, V>::ld(x + r * c + tx * V, xv);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < V; ++k) { s0[k] += xv[k]; s1[k] += xv[k] * xv[k]; }
      } else {
        float g[V];
        vecv<T, V>::ld(dy + r * c + tx * V, g);
        if (relu && beta) {   // ReLU mask recomputed from x (the bn_apply expression) instead of reading y: one tensor less
#pragma unroll
          for (int k = 0; k < V; ++k) g[k] = ((xv[k] - mu[k]) * is[k] * ga[k] + be[k]) > 0.f ? g[k] : 0.f;
        } else if (relu) {
          float yv[V];
          vecv<T, V>::ld(y + r * c + tx * V, yv);
#pragma unroll
          for (int k = 0; k < V; ++k) g[k] = yv[k] > 0.f ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < V; ++k) { s0[k] += g[k]; s1[k] += g[k] * ((xv[k] - mu[k]) * is[k]); }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { red[0][threadIdx.x][k] = s0[k]; red[1][threadIdx.x][k] = s1[k]; }
  __syncthreads();
  if (threadIdx.x < ct) {
    float a[V], b[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { a[k] = 0.f; b[k] = 0.f; }
    for (int q = 0; q < ty_n; ++q)
#pragma unroll
      for (int k = 0; k < V; ++k) { a[k] += red[0][q * ct + threadIdx.x][k]; b[k] += red[1][q * ct + threadIdx.x][k]; }
    float *out = partial + (long long)blockIdx.x * 2 * c + coff;
#pragma unroll
    for (int k = 0; k < V; k += 4) {
      *reinterpret_cast<f4 *>(out + threadIdx.x * V + k) = f4{a[k], a[k + 1], a[k + 2], a[k + 3]};
      *reinterpret_cast<f4 *>(out + c + threadIdx.x * V + k) = f4{b[k], b[k + 1], b[k + 2], b[k + 3]};
    }
  }
}
// 8 channels per lane where the shape and the pointers allow 16-byte bf16 accesses.  MEASURED SLOWER (round 5, one MI355X, bench.py
// hbm_stages): bn_stats 64@512000 23.1 vs 21.3 us, bn_backward 256@64000 52.4 vs 39.2 us -- half the row lanes per block and twice the
// accumulator registers per lane cost more than the wider loads save; these reductions are bound by their two-launch structure
// (partials + finish) and the ~60-row slabs, not by load width.  Kept as a tools switch (nrpn_set_bn_reduce_v8), OFF by default.
static std::atomic<int> g_chan_v8{0};
extern "C" int nrpn_set_bn_reduce_v8(int on) { g_chan_v8 = on ? 1 : 0; return NRPN_OK; }
static inline bool chan_v8(int dtype, int c, const void *a, const void *b, const void *d) {
  const int tile = c > 1024 ? 1024 : c;
  return dtype == NRPN_BF16 && c % 8 == 0 && tile % 8 == 0 && 256 % (tile / 8) == 0 &&
         ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

// finalize: block = (16 channels, 64 partial-lanes); fp64 accumulation of the fp32 block partials in a fixed order (lane l takes rows l, l + 64,
// ...; the 64 lane sums are added in lane order in two levels).  Round 5: 16 channels per block instead of 64 -- these kernels are pure
// latency (500-1000 partial rows of 2 C floats read by C / 64 = 1-8 workgroups took 7.7 us, 34 times per step); four times the workgroups
// and four times the lanes per channel.
constexpr int kFinCh = 16, kFinLanes = 64;
__device__ __forceinline__ void reduce_partials(const float *__restrict__ partial, int nblocks, int c, int ch, double &s, double &q) {
  __shared__ double rs[kFinLanes][kFinCh], rq[kFinLanes][kFinCh];
  __shared__ double rs2[4][kFinCh], rq2[4][kFinCh];
  double a = 0.0, b = 0.0;
  if (ch < c) {
    // two independent chains per thread (the loop is latency-, not bandwidth-bound)
    double a2[2] = {0.0, 0.0}, b2[2] = {0.0, 0.0};
    int blk = threadIdx.y;
    for (; blk + kFinLanes < nblocks; blk += 2 * kFinLanes) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        a2[u] += (double)partial[(long long)(blk + kFinLanes * u) * 2 * c + ch];
        b2[u] += (double)partial[(long long)(blk + kFinLanes * u) * 2 * c + c + ch];
      }
    }
    for (; blk < nblocks; blk += kFinLanes) {
      a2[0] += (double)partial[(long long)blk * 2 * c + ch];
      b2[0] += (double)partial[(long long)blk * 2 * c + c + ch];
    }
    a = a2[0] + a2[1];
    b = b2[0] + b2[1];
  }
  rs[threadIdx.y][threadIdx.x] = a;
  rq[threadIdx.y][threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.y < 4) {
    double t = 0.0, u = 0.0;
    for (int k = 0; k < kFinLanes / 4; ++k) { t += rs[threadIdx.y * (kFinLanes / 4) + k][threadIdx.x]; u += rq[threadIdx.y * (kFinLanes / 4) + k][threadIdx.x]; }
    rs2[threadIdx.y][threadIdx.x] = t;
    rq2[threadIdx.y][threadIdx.x] = u;
  }
  __syncthreads();
  s = 0.0; q = 0.0;
  if (threadIdx.y == 0)
    for (int k = 0; k < 4; ++k) { s += rs2[k][threadIdx.x]; q += rq2[k][threadIdx.x]; }
}

__global__ void bn_stats_finalize_kernel(const float *__restrict__ partial, int nblocks, long long rows, int c, float *__restrict__ mean,
                                         float *__restrict__ var, float *__restrict__ rmean, float *__restrict__ rvar, float momentum) {
  const int ch = blockIdx.x * kFinCh + threadIdx.x;
  double s, q;
  reduce_partials(partial, nblocks, c, ch, s, q);
  if (threadIdx.y != 0 || ch >= c) return;
  const double m = s / (double)rows;
  double v = q / (double)rows - m * m;
  if (v < 0.0) v = 0.0;
  mean[ch] = (float)m;
  var[ch] = (float)v;
  if (rmean) {
    const double unbiased = rows > 1 ? v * (double)rows / (double)(rows - 1) : v;
    rmean[ch] = (float)((1.0 - momentum) * (double)rmean[ch] + momentum * m);
    rvar[ch] = (float)((1.0 - momentum) * (double)rvar[ch] + momentum * unbiased);
  }
}

__global__ void bn_bwd_finalize_kernel(const float *__restrict__ partial, int nblocks, int c, float *__restrict__ dbeta,
                                       float *__restrict__ dgamma, float *__restrict__ acc_beta, float *__restrict__ acc_gamma) {
  const int ch = blockIdx.x * kFinCh + threadIdx.x;
  double s, q;
  reduce_partials(partial, nblocks, c, ch, s, q);
  if (threadIdx.y != 0 || ch >= c) return;
  dbeta[ch] = (float)s;
  dgamma[ch] = (float)q;
  if (acc_beta) acc_beta[ch] += (float)s;      // optional: accumulate straight into the flat gradient arena
  if (acc_gamma) acc_gamma[ch] += (float)q;
}

extern "C" size_t nrpn_bn_workspace_bytes(int64_t rows, int c) { return (size_t)(cdiv64(rows, slab_rows(rows)) * 2 * c * 4); }

static int check_bn_shape(const char *who, int64_t rows, int c) {
  const int tile = c > 1024 ? 1024 : c;
  if (rows <= 0 || c <= 0 || c % 4 != 0 || c % tile != 0 || 256 % (tile / 4) != 0)
    return nrpn_fail(NRPN_ERR_ARG, "%s: C=%d must be a multiple of 4 with min(C,1024)/4 dividing 256 and C %% 1024 == 0 above 1024 (rows=%lld)", who, c, (long long)rows);
  return 0;
}

extern "C" int nrpn_bn_stats(const void *x, int64_t rows, int c, int dtype, float *mean, float *var, float *running_mean,
                             float *running_var, float momentum, void *workspace, nrpn_stream_t stream) {
  if (int rc = check_bn_shape("bn_stats", rows, c)) return rc;
  NRPN_REQUIRE(x && mean && var && workspace, "bn_stats: null pointer");
  const int kSlab = slab_rows(rows);
  const int nb = (int)cdiv64(rows, kSlab);
  hipStream_t st = as_stream(stream);
  if (g_chan_v8.load(std::memory_order_relaxed) && chan_v8(dtype, c, x, nullptr, nullptr)) {
    hipLaunchKernelGGL((chan_partial_kernel<bf16s, 0, 8>), dim3(nb, (c + 1023) / 1024), dim3(256), 0, st, (const bf16s *)x, (const bf16s *)nullptr,
                       (const bf16s *)nullptr, (long long)rows, c, (const float *)nullptr, (const float *)nullptr, 0.f, 0, (float *)workspace, kSlab);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((chan_partial_kernel<T, 0, 4>), dim3(nb, (c + 1023) / 1024), dim3(256), 0, st, (const T *)x, (const T *)nullptr,
                                         (const T *)nullptr, (long long)rows, c, (const float *)nullptr, (const float *)nullptr, 0.f, 0,
                                         (float *)workspace, kSlab));
  }
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinCh, kFinLanes), 0, st, (const float *)workspace, nb, (long long)rows, c, mean,
                     var, running_mean, running_var, momentum);
  NRPN_LAUNCH_CHECK("bn_stats");
  return NRPN_OK;
}

// Finish partial statistics produced elsewhere ([nparts][2][C] = per-part (sum, sum of squares) over disjoint row sets, e.g. by the conv
// epilogue: nrpn_conv3d_fwd_stats) into mean / biased variance (+ running statistics), exactly as nrpn_bn_stats finishes its own slabs.
extern "C" int nrpn_bn_stats_finalize(const float *partials, int nparts, int64_t rows, int c, float *mean, float *var, float *running_mean,
                                      float *running_var, float momentum, nrpn_stream_t stream) {
  NRPN_REQUIRE(partials && mean && var && nparts > 0 && rows > 0 && c > 0, "bn_stats_finalize: bad arguments");
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinCh, kFinLanes), 0, as_stream(stream), partials, nparts, (long long)rows, c,
                     mean, var, running_mean, running_var, momentum);
  NRPN_LAUNCH_CHECK("bn_stats_finalize");
  return NRPN_OK;
}

// y = relu?((x - mean) * rsqrt(var + eps) * gamma + beta)   -- grid-stride over 4-channel groups
template <typename T>
__global__ void bn_apply_kernel(const T *__restrict__ x, T *__restrict__ y, long long groups, int c, const float *__restrict__ mean,
                                const float *__restrict__ var, const float *__restrict__ gamma, const float *__restrict__ beta, float eps,
                                int relu) {
  const int ct = c / 4;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(g % ct) * 4;
    const f4 xv = vec4<T>::ld(x + g * 4);
    const f4 mu = *reinterpret_cast<const f4 *>(mean + cg), vv = *reinterpret_cast<const f4 *>(var + cg);
    const f4 ga = *reinterpret_cast<const f4 *>(gamma + cg), be = *reinterpret_cast<const f4 *>(beta + cg);
    f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float is = 1.0f / sqrtf(vv[k] + eps);
      o[k] = (xv[k] - mu[k]) * is * ga[k] + be[k];
      if (relu) o[k] = fmaxf(o[k], 0.f);
    }
    vec4<T>::st(y + g * 4, o);
  }
}

static inline int ew_blocks(long long work) { long long b = (work + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// Fast forms of bn_apply / bn_bwd_apply.  V channels per lane in 16-byte accesses (8 bf16 / 4 fp32) and a grid stride that is a multiple
// of C / V: a lane then meets the SAME channels in every iteration, so the per-channel parameters, the rsqrt and the 64-bit "g % (C / V)"
// leave the loop, whose body is the loads plus a handful of VALU operations per element.  These kernels run next to the weight-gradient
// stream's MFMA kernels (3x their stand-alone time there): every VALU slot they take is one the MFMA waves wait for.  The arithmetic is
// the expression of the general kernels, term for term (the ReLU mask recomputed in backward has to agree with what forward wrote).
// Launch condition (bn_fast_v): 256 % (C / V) == 0, i.e. C / V a power of two <= 256 -- every BatchNorm of the three backbones.
static std::atomic<int> g_bn_fast{1};      // tools-only A/B switch (nerfrpn_tools.h)
extern "C" int nrpn_set_bn_fast(int on) { g_bn_fast = on ? 1 : 0; return NRPN_OK; }
static inline int bn_fast_v(int c, int dtype) {
  if (!g_bn_fast) return 0;
  const int v = (dtype != NRPN_F32 && c % 8 == 0) ? 8 : 4;
  const int ctv = c / v;
  return (c % v == 0 && ctv > 0 && ctv <= 256 && 256 % ctv == 0) ? v : 0;
}

// V per-channel fp32 parameters as 16-byte loads (cg is a multiple of V >= 4 floats)
template <int V>
__device__ __forceinline__ void ld_params(const float *__restrict__ p, float *o) {
#pragma unroll
  for (int q = 0; q < V; q += 4) {
    const f4 v = *reinterpret_cast<const f4 *>(p + q);
    o[q] = v[0]; o[q + 1] = v[1]; o[q + 2] = v[2]; o[q + 3] = v[3];
  }
}

// grid of the fast apply kernels: one group per lane up to one resident round of the chip (256 CUs x `per_cu` blocks at the kernel's register
// count), beyond that more groups per lane -- every lane pays a prologue (parameters, V rsqrt), and a second round of blocks would only add a
// tail.  The stride blocks * 256 is a multiple of every power of two <= 256.
static inline int bn_fast_blocks(long long groups, int per_cu) {
  const long long b = (groups + 255) / 256, cap = 256ll * per_cu;
  return (int)(b > cap ? cap : (b < 1 ? 1 : b));
}

template <typename T, int V>
__global__ void __launch_bounds__(256) bn_apply_fast_kernel(const T *__restrict__ x, T *__restrict__ y, long long groups, int c,
                                                            const float *__restrict__ mean, const float *__restrict__ var,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta, float eps, int relu) {
  const long long stride = (long long)gridDim.x * 256;
  long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const int cg = ((int)g & (c / V - 1)) * V;              // C / V is a power of two dividing the stride: loop invariant
  float mu[V], is[V], ga[V], be[V];
  ld_params<V>(mean + cg, mu); ld_params<V>(var + cg, is); ld_params<V>(gamma + cg, ga); ld_params<V>(beta + cg, be);
#pragma unroll
  for (int k = 0; k < V; ++k) is[k] = 1.0f / sqrtf(is[k] + eps);
#pragma unroll 2
  for (; g < groups; g += stride) {
    float xv[V], o[V];
    vecv<T, V>::ld(x + g * V, xv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      o[k] = (xv[k] - mu[k]) * is[k] * ga[k] + be[k];
      if (relu) o[k] = fmaxf(o[k], 0.f);
    }
    vecv<T, V>::st(y + g * V, o);
  }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_fast_kernel(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy,
                                                                T *__restrict__ dx, long long groups, int c, long long rows,
                                                                const float *__restrict__ mean, const float *__restrict__ var,
                                                                const float *__restrict__ gamma, float eps, int relu,
                                                                const float *__restrict__ dgamma, const float *__restrict__ dbeta,
                                                                const float *__restrict__ beta) {
  const float invr = 1.0f / (float)rows;
  const long long stride = (long long)gridDim.x * 256;
  long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  const int cg = ((int)g & (c / V - 1)) * V;
  float mu[V], is[V], ga[V], dg[V], db[V], be[V];
  ld_params<V>(mean + cg, mu); ld_params<V>(var + cg, is); ld_params<V>(gamma + cg, ga); ld_params<V>(dgamma + cg, dg); ld_params<V>(dbeta + cg, db);
#pragma unroll
  for (int k = 0; k < V; ++k) { is[k] = 1.0f / sqrtf(is[k] + eps); be[k] = 0.f; }
  if (relu && beta) ld_params<V>(beta + cg, be);
#pragma unroll 2
  for (; g < groups; g += stride) {
    float xv[V], gv[V], o[V];
    vecv<T, V>::ld(x + g * V, xv);
    vecv<T, V>::ld(dy + g * V, gv);
    if (relu && !beta) {
      float yv[V];
      vecv<T, V>::ld(y + g * V, yv);
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if (relu && beta && !(((xv[k] - mu[k]) * is[k] * ga[k] + be[k]) > 0.f)) gv[k] = 0.f;    // mask recomputed from x, y is not read
      const float xh = (xv[k] - mu[k]) * is[k];
      o[k] = ga[k] * is[k] * (gv[k] - db[k] * invr - xh * dg[k] * invr);
    }
    vecv<T, V>::st(dx + g * V, o);
  }
}

extern "C" int nrpn_bn_apply(const void *x, void *y, int64_t rows, int c, int dtype, const float *mean, const float *var,
                             const float *gamma, const float *beta, float eps, int relu, nrpn_stream_t stream) {
  NRPN_REQUIRE(rows > 0 && c > 0 && c % 4 == 0, "bn_apply: C=%d must be a multiple of 4", c);
  NRPN_REQUIRE(x && y && mean && var && gamma && beta, "bn_apply: null pointer");
  int fv = bn_fast_v(c, dtype);
  if (fv == 8 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)) fv = 0;      // 16-byte accesses
  const long long groups = rows * (c / (fv ? fv : 4));
  if (fv == 8)
    hipLaunchKernelGGL((bn_apply_fast_kernel<bf16s, 8>), dim3(bn_fast_blocks(groups, 8)), dim3(256), 0, as_stream(stream), (const bf16s *)x, (bf16s *)y,
                       groups, c, mean, var, gamma, beta, eps, relu);
  else if (fv == 4) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply_fast_kernel<T, 4>), dim3(bn_fast_blocks(groups, 8)), dim3(256), 0, as_stream(stream), (const T *)x,
                                         (T *)y, groups, c, mean, var, gamma, beta, eps, relu));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(ew_blocks(groups)), dim3(256), 0, as_stream(stream), (const T *)x, (T *)y,
                                         groups, c, mean, var, gamma, beta, eps, relu));
  }
  NRPN_LAUNCH_CHECK("bn_apply");
  return NRPN_OK;
}

// dx = gamma * invstd * (dy' - dbeta / R - xhat * dgamma / R)
template <typename T>
__global__ void bn_bwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ y, const T *__restrict__ dy, T *__restrict__ dx,
                                    long long groups, int c, long long rows, const float *__restrict__ mean, const float *__restrict__ var,
                                    const float *__restrict__ gamma, float eps, int relu, const float *__restrict__ dgamma,
                                    const float *__restrict__ dbeta, const float *__restrict__ beta) {
  const int ct = c / 4;
  const float invr = 1.0f / (float)rows;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(g % ct) * 4;
    const f4 xv = vec4<T>::ld(x + g * 4);
    f4 gv = vec4<T>::ld(dy + g * 4);
    const f4 mu = *reinterpret_cast<const f4 *>(mean + cg), vv = *reinterpret_cast<const f4 *>(var + cg);
    const f4 ga = *reinterpret_cast<const f4 *>(gamma + cg);
    const f4 dg = *reinterpret_cast<const f4 *>(dgamma + cg), db = *reinterpret_cast<const f4 *>(dbeta + cg);
    if (relu && !beta) {
      const f4 yv = vec4<T>::ld(y + g * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    }
    f4 be = {0, 0, 0, 0};
    if (relu && beta) be = *reinterpret_cast<const f4 *>(beta + cg);
    f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float is = 1.0f / sqrtf(vv[k] + eps);
      if (relu && beta && !(((xv[k] - mu[k]) * is * ga[k] + be[k]) > 0.f)) gv[k] = 0.f;    // mask recomputed from x, y is not read
      const float xh = (xv[k] - mu[k]) * is;
      o[k] = ga[k] * is * (gv[k] - db[k] * invr - xh * dg[k] * invr);
    }
    vec4<T>::st(dx + g * 4, o);
  }
}

extern "C" int nrpn_bn_backward(const void *x, const void *y, const void *dy, void *dx, int64_t rows, int c, int dtype, const float *mean,
                                const float *var, const float *gamma, const float *beta, float eps, int relu, float *dgamma, float *dbeta,
                                float *acc_dgamma, float *acc_dbeta, void *workspace, nrpn_stream_t stream) {
  if (int rc = check_bn_shape("bn_backward", rows, c)) return rc;
  NRPN_REQUIRE(x && dy && dx && mean && var && gamma && dgamma && dbeta && workspace && (!relu || y || beta), "bn_backward: null pointer");
  const int kSlab = slab_rows(rows);
  const int nb = (int)cdiv64(rows, kSlab);
  hipStream_t st = as_stream(stream);
  if (g_chan_v8.load(std::memory_order_relaxed) && chan_v8(dtype, c, x, y, dy)) {
    hipLaunchKernelGGL((chan_partial_kernel<bf16s, 1, 8>), dim3(nb, (c + 1023) / 1024), dim3(256), 0, st, (const bf16s *)x, (const bf16s *)y,
                       (const bf16s *)dy, (long long)rows, c, mean, var, eps, relu, (float *)workspace, kSlab, gamma, beta);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((chan_partial_kernel<T, 1, 4>), dim3(nb, (c + 1023) / 1024), dim3(256), 0, st, (const T *)x, (const T *)y, (const T *)dy,
                                         (long long)rows, c, mean, var, eps, relu, (float *)workspace, kSlab, gamma, beta));
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((c + kFinCh - 1) / kFinCh), dim3(kFinCh, kFinLanes), 0, st, (const float *)workspace, nb, c, dbeta, dgamma, acc_dbeta, acc_dgamma);
  int fv = bn_fast_v(c, dtype);
  if (fv == 8 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15))
    fv = 0;                                                                                                // 16-byte accesses
  const long long groups = rows * (c / (fv ? fv : 4));
  if (fv == 8)
    hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<bf16s, 8>), dim3(bn_fast_blocks(groups, 5)), dim3(256), 0, st, (const bf16s *)x, (const bf16s *)y,
                       (const bf16s *)dy, (bf16s *)dx, groups, c, (long long)rows, mean, var, gamma, eps, relu, (const float *)dgamma,
                       (const float *)dbeta, beta);
  else if (fv == 4) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<T, 4>), dim3(bn_fast_blocks(groups, 8)), dim3(256), 0, st, (const T *)x, (const T *)y,
                                         (const T *)dy, (T *)dx, groups, c, (long long)rows, mean, var, gamma, eps, relu,
                                         (const float *)dgamma, (const float *)dbeta, beta));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(ew_blocks(groups)), dim3(256), 0, st, (const T *)x, (const T *)y,
                                         (const T *)dy, (T *)dx, groups, c, (long long)rows, mean, var, gamma, eps, relu,
                                         (const float *)dgamma, (const float *)dbeta, beta));
  }
  NRPN_LAUNCH_CHECK("bn_backward");
  return NRPN_OK;
}

template <typename T>
__global__ void relu_bwd_kernel(const T *__restrict__ y, const T *__restrict__ dy, T *__restrict__ dx, long long groups) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long long)gridDim.x * blockDim.x) {
    const f4 yv = vec4<T>::ld(y + g * 4);
    f4 gv = vec4<T>::ld(dy + g * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    vec4<T>::st(dx + g * 4, gv);
  }
}

extern "C" int nrpn_relu_backward(const void *y, const void *dy, void *dx, int64_t count, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(count > 0 && count % 4 == 0, "relu_backward: count must be a positive multiple of 4");
  NRPN_REQUIRE(y && dy && dx, "relu_backward: null pointer");
  const long long groups = count / 4;
  DISPATCH_T(dtype, hipLaunchKernelGGL(relu_bwd_kernel<T>, dim3(ew_blocks(groups)), dim3(256), 0, as_stream(stream), (const T *)y,
                                       (const T *)dy, (T *)dx, groups));
  NRPN_LAUNCH_CHECK("relu_backward");
  return NRPN_OK;
}

// =====================================================================================================================
// MaxPool3d (torch semantics: -inf padding, ceil_mode windows must start inside input+left pad, first max wins)
// =====================================================================================================================
static inline int pool_out(int in, int k, int s, int p, int ceil_mode) {
  int o = ceil_mode ? (in + 2 * p - k + s - 1) / s + 1 : (in + 2 * p - k) / s + 1;
  if (ceil_mode && (o - 1) * s >= in + p) --o;
  return o;
}
extern "C" int nrpn_pool_out_size(int in, int k, int s, int p, int ceil_mode) { return pool_out(in, k, s, p, ceil_mode); }

// I = index type of the flat group / voxel arithmetic: unsigned (one 32-bit division per axis) whenever the tensors hold < 2^31 elements --
// the 64-bit divisions of the first version were ~4x the VALU work of everything else in these kernels and bound them at a third of the HBM
// rate (round 5) -- long long otherwise.
template <int V> struct argpack;
template <> struct argpack<4> { typedef unsigned int type; };
template <> struct argpack<8> { typedef unsigned long long type; };

template <typename T, int V, typename I>
__global__ void maxpool_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, int8_t *__restrict__ arg, int n, int gx, int gy, int gz, int ox,
                                   int oy, int oz, int c, int k, int s, int p) {
  const I ct = (I)(c / V);
  const I total = (I)n * ox * oy * oz * ct;
  for (I g = (I)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (I)gridDim.x * blockDim.x) {
    const I vox = g / ct;
    const int cg = (int)(g - vox * ct) * V;
    I v = vox;
    const int z = (int)(v % (I)oz); v /= (I)oz;
    const int yy = (int)(v % (I)oy); v /= (I)oy;
    const int xx = (int)(v % (I)ox);
    const long long b = (long long)(v / (I)ox);
    float best[V];
    int bi[V];
    bool first = true;
    // window clipped to the grid once per axis (torch: -inf padding, first maximum wins => ascending scan order is kept)
    const int a0 = max(0, p - xx * s), a1 = min(k, gx + p - xx * s);
    const int b0 = max(0, p - yy * s), b1 = min(k, gy + p - yy * s);
    const int d0 = max(0, p - z * s), d1 = min(k, gz + p - z * s);
    for (int a = a0; a < a1; ++a) {
      const int ix = xx * s - p + a;
      for (int bq = b0; bq < b1; ++bq) {
        const int iy = yy * s - p + bq;
        for (int d = d0; d < d1; ++d) {
          const int iz = z * s - p + d;
          float xv[V];
          vecv<T, V>::ld(x + ((((b * gx + ix) * gy + iy) * gz + iz) * (long long)c + cg), xv);
          const int code = (a * k + bq) * k + d;
#pragma unroll
          for (int q = 0; q < V; ++q)
            if (first || xv[q] > best[q]) { best[q] = xv[q]; bi[q] = code; }
          first = false;
        }
      }
    }
    if (first) {
#pragma unroll
      for (int q = 0; q < V; ++q) { best[q] = -INFINITY; bi[q] = 0; }
    }
    const long long o = (long long)vox * c + cg;
    vecv<T, V>::st(y + o, best);
    if (arg) {           // V argmax codes as one 4- / 8-byte store
      typename argpack<V>::type pk = 0;
#pragma unroll
      for (int q = 0; q < V; ++q) pk |= (typename argpack<V>::type)(unsigned char)bi[q] << (8 * q);
      *reinterpret_cast<typename argpack<V>::type *>(arg + o) = pk;
    }
  }
}

// Round 6: the 32-bit forms above still spent most of their time dividing -- four runtime udivs per lane for the voxel decomposition and, in
// the backward, a runtime `% s` and `/ s` per axis and per candidate window: ~200 VALU instructions per lane against 8 - 27 loads, VALU-bound at
// 0.18 - 0.31 of the HBM rate.  The fast forms take the divisors as multiply-shift pairs (common.h FastDiv) and the stride as
// a template argument (1 or 2: shifts and masks); results are those of the general kernels, bit for bit (same scan order, same first-maximum rule).
struct PoolDiv { FastDiv ct, z, y, x; };
static std::atomic<int> g_pool_fast{1};      // tools-only A/B switch (nerfrpn_tools.h): 0 = the general index arithmetic (pools, upsample-add)
extern "C" int nrpn_set_pool_fast(int on) { g_pool_fast = on ? 1 : 0; return NRPN_OK; }

// K = window size as a template argument (2 or 3; 0 = runtime loops): every load of a lane is then issued before the first comparison -- the
// runtime-bounded loops with their data-dependent compare chain serialised 27 load latencies per lane (31 us for the 3/2/1 pool of the stem).
// Out-of-grid window positions read a clamped address and are masked out of the comparison; `first` keeps torch's rule (the first in-grid
// element is taken whatever its value, later ones only if strictly greater): same bits as the general kernel.
template <typename T, int V, int K>
__global__ void __launch_bounds__(256) maxpool_fwd_fast_kernel(const T *__restrict__ x, T *__restrict__ y, int8_t *__restrict__ arg, unsigned total,
                                                               int gx, int gy, int gz, int ox, int oy, int oz, int c, int k, int s, int p, PoolDiv dv) {
  for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
    const unsigned vox = fastdiv(g, dv.ct);
    const int cg = (int)(g - vox * dv.ct.d) * V;
    const unsigned t1 = fastdiv(vox, dv.z), t2 = fastdiv(t1, dv.y), b = fastdiv(t2, dv.x);
    const int z = (int)(vox - t1 * dv.z.d), yy = (int)(t1 - t2 * dv.y.d), xx = (int)(t2 - b * dv.x.d);
    float best[V];
    int bi[V];
    bool first = true;
    const T *xb = x + (long long)b * gx * gy * gz * c + cg;
    if constexpr (K > 0) {
      const int x0 = xx * s - p, y0 = yy * s - p, z0 = z * s - p;
      float xv[K * K * K][V];
#pragma unroll
      for (int a = 0; a < K; ++a)
#pragma unroll
        for (int bq = 0; bq < K; ++bq)
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const int ix = min(max(x0 + a, 0), gx - 1), iy = min(max(y0 + bq, 0), gy - 1), iz = min(max(z0 + d, 0), gz - 1);
            vecv<T, V>::ld(xb + (long long)((ix * gy + iy) * gz + iz) * c, xv[(a * K + bq) * K + d]);
          }
#pragma unroll
      for (int q = 0; q < V; ++q) { best[q] = -INFINITY; bi[q] = 0; }
#pragma unroll
      for (int a = 0; a < K; ++a)
#pragma unroll
        for (int bq = 0; bq < K; ++bq)
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const bool in = (unsigned)(x0 + a) < (unsigned)gx && (unsigned)(y0 + bq) < (unsigned)gy && (unsigned)(z0 + d) < (unsigned)gz;
            const int code = (a * K + bq) * K + d;
#pragma unroll
            for (int q = 0; q < V; ++q)
              if (in && (first || xv[code][q] > best[q])) { best[q] = xv[code][q]; bi[q] = code; }
            first = first && !in;
          }
    } else {
      const int a0 = max(0, p - xx * s), a1 = min(k, gx + p - xx * s);
      const int b0 = max(0, p - yy * s), b1 = min(k, gy + p - yy * s);
      const int d0 = max(0, p - z * s), d1 = min(k, gz + p - z * s);
      for (int a = a0; a < a1; ++a) {
        const int ix = xx * s - p + a;
        for (int bq = b0; bq < b1; ++bq) {
          const int iy = yy * s - p + bq;
          const T *row = xb + (long long)((ix * gy + iy) * gz + (z * s - p)) * c;
          const int code0 = (a * k + bq) * k;
          if (k == 3) {       // the z run of a 3-wide window as three loads in flight (clamped address, masked compare): 9 latencies per lane, not 27
            float xv[3][V];
#pragma unroll
            for (int d = 0; d < 3; ++d) vecv<T, V>::ld(row + (long long)min(max(d, d0), d1 - 1) * c, xv[d]);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              const bool in = d >= d0 && d < d1;
#pragma unroll
              for (int q = 0; q < V; ++q)
                if (in && (first || xv[d][q] > best[q])) { best[q] = xv[d][q]; bi[q] = code0 + d; }
              first = first && !in;
            }
            continue;
          }
          for (int d = d0; d < d1; ++d) {
            float xv[V];
            vecv<T, V>::ld(row + (long long)d * c, xv);
#pragma unroll
            for (int q = 0; q < V; ++q)
              if (first || xv[q] > best[q]) { best[q] = xv[q]; bi[q] = code0 + d; }
            first = false;
          }
        }
      }
      if (first) {
#pragma unroll
        for (int q = 0; q < V; ++q) { best[q] = -INFINITY; bi[q] = 0; }
      }
    }
    const long long o = (long long)vox * c + cg;
    vecv<T, V>::st(y + o, best);
    if (arg) {
      typename argpack<V>::type pk = 0;
#pragma unroll
      for (int q = 0; q < V; ++q) pk |= (typename argpack<V>::type)(unsigned char)bi[q] << (8 * q);
      *reinterpret_cast<typename argpack<V>::type *>(arg + o) = pk;
    }
  }
}

// S = stride (1 or 2), K = window size (2 or 3: at most NC = ceil(K / S) candidate windows per axis, all NC^3 loads issued up front;
// 0 = runtime loops).  Per axis the candidate windows of coordinate x are those with offset a = (x + p) mod S, + S, ... < k.
template <typename T, int V, int S, int K>
__global__ void __launch_bounds__(256) maxpool_bwd_fast_kernel(const T *__restrict__ dy, const int8_t *__restrict__ arg, T *__restrict__ dx, unsigned total,
                                                               int ox, int oy, int oz, int c, int k, int p, PoolDiv dv) {
  for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
    const unsigned vox = fastdiv(g, dv.ct);
    const int cg = (int)(g - vox * dv.ct.d) * V;
    const unsigned t1 = fastdiv(vox, dv.z), t2 = fastdiv(t1, dv.y), b = fastdiv(t2, dv.x);
    const int z = (int)(vox - t1 * dv.z.d), yy = (int)(t1 - t2 * dv.y.d), xx = (int)(t2 - b * dv.x.d);
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    const long long ob = (long long)b * ox * oy * oz;
    if constexpr (K > 0) {
      constexpr int NC = (K + S - 1) / S;
      const int ra = (xx + p) & (S - 1), rb = (yy + p) & (S - 1), rd = (z + p) & (S - 1);
      float gv[NC * NC * NC][V];
      typename argpack<V>::type pk[NC * NC * NC];
#pragma unroll
      for (int ja = 0; ja < NC; ++ja)
#pragma unroll
        for (int jb = 0; jb < NC; ++jb)
#pragma unroll
          for (int jd = 0; jd < NC; ++jd) {
            const int tx = xx + p - (ra + S * ja), ty = yy + p - (rb + S * jb), tz = z + p - (rd + S * jd);
            const int wx = min(max(S == 2 ? tx >> 1 : tx, 0), ox - 1), wy = min(max(S == 2 ? ty >> 1 : ty, 0), oy - 1),
                      wz = min(max(S == 2 ? tz >> 1 : tz, 0), oz - 1);
            const long long o = (ob + (long long)(wx * oy + wy) * oz + wz) * c + cg;
            vecv<T, V>::ld(dy + o, gv[(ja * NC + jb) * NC + jd]);
            pk[(ja * NC + jb) * NC + jd] = *reinterpret_cast<const typename argpack<V>::type *>(arg + o);
          }
      // the general kernel adds the candidates in ascending (a, b, d) order: keep it (fp32 sums of up to 8 terms)
#pragma unroll
      for (int ja = 0; ja < NC; ++ja)
#pragma unroll
        for (int jb = 0; jb < NC; ++jb)
#pragma unroll
          for (int jd = 0; jd < NC; ++jd) {
            const int a = ra + S * ja, bq = rb + S * jb, d = rd + S * jd;
            const int tx = xx + p - a, ty = yy + p - bq, tz = z + p - d;
            const bool ok = a < K && bq < K && d < K && tx >= 0 && ty >= 0 && tz >= 0 && (S == 2 ? tx >> 1 : tx) < ox &&
                            (S == 2 ? ty >> 1 : ty) < oy && (S == 2 ? tz >> 1 : tz) < oz;
            const int code = (a * K + bq) * K + d;
            const int idx = (ja * NC + jb) * NC + jd;
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += (ok && (int)((pk[idx] >> (8 * q)) & 0xff) == code) ? gv[idx][q] : 0.f;
          }
    } else {
      for (int a = (xx + p) & (S - 1); a < k; a += S) {
        const int tx = xx + p - a, wx = S == 2 ? tx >> 1 : tx;
        if (tx < 0 || wx >= ox) continue;
        for (int bq = (yy + p) & (S - 1); bq < k; bq += S) {
          const int ty = yy + p - bq, wy = S == 2 ? ty >> 1 : ty;
          if (ty < 0 || wy >= oy) continue;
          const long long orow = (ob + (long long)(wx * oy + wy) * oz) * c + cg;
          const int code0 = (a * k + bq) * k;
          for (int d = (z + p) & (S - 1); d < k; d += S) {
            const int tz = z + p - d, wz = S == 2 ? tz >> 1 : tz;
            if (tz < 0 || wz >= oz) continue;
            const long long o = orow + (long long)wz * c;
            float gvv[V];
            vecv<T, V>::ld(dy + o, gvv);
            const typename argpack<V>::type pkk = *reinterpret_cast<const typename argpack<V>::type *>(arg + o);
            const int code = code0 + d;
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += ((int)((pkk >> (8 * q)) & 0xff) == code) ? gvv[q] : 0.f;
          }
        }
      }
    }
    vecv<T, V>::st(dx + (long long)vox * c + cg, acc);
  }
}

// gather form: every input voxel sums the dy of the windows whose argmax points at it (no atomics, deterministic).  Per axis the
// windows containing coordinate x are those with offset a = (x + p) mod s, + s, + 2s, ... < k: at most ceil(k / s) candidates,
// enumerated directly (8 for the 3/2/1 pool instead of testing all 27 offsets).
template <typename T, int V, typename I>
__global__ void maxpool_bwd_kernel(const T *__restrict__ dy, const int8_t *__restrict__ arg, T *__restrict__ dx, int n, int gx, int gy, int gz,
                                   int ox, int oy, int oz, int c, int k, int s, int p) {
  const I ct = (I)(c / V);
  const I total = (I)n * gx * gy * gz * ct;
  for (I g = (I)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (I)gridDim.x * blockDim.x) {
    const I vox = g / ct;
    const int cg = (int)(g - vox * ct) * V;
    I v = vox;
    const int z = (int)(v % (I)gz); v /= (I)gz;
    const int yy = (int)(v % (I)gy); v /= (I)gy;
    const int xx = (int)(v % (I)gx);
    const long long b = (long long)(v / (I)gx);
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    for (int a = (xx + p) % s; a < k; a += s) {
      const int wx = (xx + p - a) / s;
      if (xx + p - a < 0 || wx >= ox) continue;
      for (int bq = (yy + p) % s; bq < k; bq += s) {
        const int wy = (yy + p - bq) / s;
        if (yy + p - bq < 0 || wy >= oy) continue;
        for (int d = (z + p) % s; d < k; d += s) {
          const int wz = (z + p - d) / s;
          if (z + p - d < 0 || wz >= oz) continue;
          const long long o = ((((b * ox + wx) * oy + wy) * oz + wz) * (long long)c + cg);
          const int code = (a * k + bq) * k + d;
          float gv[V];
          vecv<T, V>::ld(dy + o, gv);
          const typename argpack<V>::type pk = *reinterpret_cast<const typename argpack<V>::type *>(arg + o);      // V codes in one load
#pragma unroll
          for (int q = 0; q < V; ++q) acc[q] += ((int)((pk >> (8 * q)) & 0xff) == code) ? gv[q] : 0.f;
        }
      }
    }
    vecv<T, V>::st(dx + (long long)vox * c + cg, acc);
  }
}

extern "C" int nrpn_maxpool3d_fwd(const void *x, void *y, int8_t *argmax, int n, int gx, int gy, int gz, int c, int k, int s, int p,
                                  int ceil_mode, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0 && c > 0 && c % 4 == 0 && k >= 1 && k <= 5 && s >= 1 && p >= 0, "maxpool_fwd: bad sizes");
  NRPN_REQUIRE(x && y, "maxpool_fwd: null pointer");
  const int ox = pool_out(gx, k, s, p, ceil_mode), oy = pool_out(gy, k, s, p, ceil_mode), oz = pool_out(gz, k, s, p, ceil_mode);
  const bool small = (long long)n * gx * gy * gz * c < (1ll << 31);      // 32-bit index arithmetic (every real shape); 64-bit beyond
#define NRPN_POOL_FWD(T_, V_, I_) hipLaunchKernelGGL((maxpool_fwd_kernel<T_, V_, I_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                     (const T_ *)x, (T_ *)y, argmax, n, gx, gy, gz, ox, oy, oz, c, k, s, p)
  if (small && g_pool_fast.load(std::memory_order_relaxed)) {
    const int v = (dtype == NRPN_BF16 && c % 8 == 0) ? 8 : 4;
    const long long total = (long long)n * ox * oy * oz * (c / v);
    const PoolDiv dv{make_fastdiv((unsigned)(c / v)), make_fastdiv((unsigned)oz), make_fastdiv((unsigned)oy), make_fastdiv((unsigned)ox)};
#define NRPN_POOL_FWDF(T_, V_, K_) hipLaunchKernelGGL((maxpool_fwd_fast_kernel<T_, V_, K_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                      (const T_ *)x, (T_ *)y, argmax, (unsigned)total, gx, gy, gz, ox, oy, oz, c, k, s, p, dv)
#define NRPN_POOL_FWDK(T_, V_) do { if (k == 2) NRPN_POOL_FWDF(T_, V_, 2); else NRPN_POOL_FWDF(T_, V_, 0); } while (0)   /* k = 3 fully unrolled: 216 registers of loads, measured slower (35.2 vs 31.5 us) */
    if (v == 8) NRPN_POOL_FWDK(bf16s, 8); else { DISPATCH_T(dtype, NRPN_POOL_FWDK(T, 4)); }
#undef NRPN_POOL_FWDK
#undef NRPN_POOL_FWDF
  } else if (dtype == NRPN_BF16 && c % 8 == 0) {
    const long long total = (long long)n * ox * oy * oz * (c / 8);
    if (small) NRPN_POOL_FWD(bf16s, 8, unsigned); else NRPN_POOL_FWD(bf16s, 8, long long);
  } else {
    const long long total = (long long)n * ox * oy * oz * (c / 4);
    if (small) { DISPATCH_T(dtype, NRPN_POOL_FWD(T, 4, unsigned)); } else { DISPATCH_T(dtype, NRPN_POOL_FWD(T, 4, long long)); }
  }
#undef NRPN_POOL_FWD
  NRPN_LAUNCH_CHECK("maxpool_fwd");
  return NRPN_OK;
}

extern "C" int nrpn_maxpool3d_bwd(const void *dy, const int8_t *argmax, void *dx, int n, int gx, int gy, int gz, int c, int k, int s, int p,
                                  int ceil_mode, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0 && c > 0 && c % 4 == 0 && k >= 1 && k <= 5 && s >= 1 && p >= 0, "maxpool_bwd: bad sizes");
  NRPN_REQUIRE(dy && argmax && dx, "maxpool_bwd: null pointer");
  const int ox = pool_out(gx, k, s, p, ceil_mode), oy = pool_out(gy, k, s, p, ceil_mode), oz = pool_out(gz, k, s, p, ceil_mode);
  const bool small = (long long)n * gx * gy * gz * c < (1ll << 31);
#define NRPN_POOL_BWD(T_, V_, I_) hipLaunchKernelGGL((maxpool_bwd_kernel<T_, V_, I_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                     (const T_ *)dy, argmax, (T_ *)dx, n, gx, gy, gz, ox, oy, oz, c, k, s, p)
  if (small && (s == 1 || s == 2) && g_pool_fast.load(std::memory_order_relaxed)) {
    const int v = (dtype == NRPN_BF16 && c % 8 == 0) ? 8 : 4;
    const long long total = (long long)n * gx * gy * gz * (c / v);
    const PoolDiv dv{make_fastdiv((unsigned)(c / v)), make_fastdiv((unsigned)gz), make_fastdiv((unsigned)gy), make_fastdiv((unsigned)gx)};
#define NRPN_POOL_BWDF(T_, V_, S_, K_) hipLaunchKernelGGL((maxpool_bwd_fast_kernel<T_, V_, S_, K_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                          (const T_ *)dy, argmax, (T_ *)dx, (unsigned)total, ox, oy, oz, c, k, p, dv)
    // stride 2: the two pools of the VGG / ResNet paths (3/2/1 and 2/2/0) with every load up front; everything else on the runtime loops
#define NRPN_POOL_BWDK(T_, V_) do { if (s == 2 && k == 2) NRPN_POOL_BWDF(T_, V_, 2, 2); /* (k = 3 unrolled loads all 8 candidates where 3.4 exist on average: 49.5 vs 35.4 us) */ \
                                    else if (s == 2) NRPN_POOL_BWDF(T_, V_, 2, 0); else NRPN_POOL_BWDF(T_, V_, 1, 0); } while (0)
    if (v == 8) NRPN_POOL_BWDK(bf16s, 8); else { DISPATCH_T(dtype, NRPN_POOL_BWDK(T, 4)); }
#undef NRPN_POOL_BWDK
#undef NRPN_POOL_BWDF
  } else if (dtype == NRPN_BF16 && c % 8 == 0) {
    const long long total = (long long)n * gx * gy * gz * (c / 8);
    if (small) NRPN_POOL_BWD(bf16s, 8, unsigned); else NRPN_POOL_BWD(bf16s, 8, long long);
  } else {
    const long long total = (long long)n * gx * gy * gz * (c / 4);
    if (small) { DISPATCH_T(dtype, NRPN_POOL_BWD(T, 4, unsigned)); } else { DISPATCH_T(dtype, NRPN_POOL_BWD(T, 4, long long)); }
  }
#undef NRPN_POOL_BWD
  NRPN_LAUNCH_CHECK("maxpool_bwd");
  return NRPN_OK;
}

// =====================================================================================================================
// FPN top-down: fine += nearest(coarse), legacy index rule src = min(floor(dst * (in/out)), in-1) in fp32
// =====================================================================================================================
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  const float scale = (float)in / (float)out;
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

template <typename T, int V, typename I>
__global__ void upsample_add_fwd_kernel(T *__restrict__ fine, const T *__restrict__ coarse, int n, int fx, int fy, int fz, int cx, int cy,
                                        int cz, int c) {
  const I ct = (I)(c / V);
  const I total = (I)n * fx * fy * fz * ct;
  for (I g = (I)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (I)gridDim.x * blockDim.x) {
    const I vox = g / ct;
    const int cg = (int)(g - vox * ct) * V;
    I v = vox;
    const int z = (int)(v % (I)fz); v /= (I)fz;
    const int y = (int)(v % (I)fy); v /= (I)fy;
    const int x = (int)(v % (I)fx);
    const long long b = (long long)(v / (I)fx);
    const long long src = (((b * cx + nearest_src(x, cx, fx)) * cy + nearest_src(y, cy, fy)) * cz + nearest_src(z, cz, fz)) * (long long)c + cg;
    const long long dst = (long long)vox * c + cg;
    float a[V], bq[V];
    vecv<T, V>::ld(fine + dst, a);
    vecv<T, V>::ld(coarse + src, bq);
#pragma unroll
    for (int k = 0; k < V; ++k) a[k] += bq[k];
    vecv<T, V>::st(fine + dst, a);
  }
}

template <typename T, int V, typename I>
__global__ void upsample_add_bwd_kernel(const T *__restrict__ dfine, T *__restrict__ dcoarse, int n, int fx, int fy, int fz, int cx, int cy,
                                        int cz, int c, int accumulate) {
  const I ct = (I)(c / V);
  const I total = (I)n * cx * cy * cz * ct;
  for (I g = (I)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (I)gridDim.x * blockDim.x) {
    const I vox = g / ct;
    const int cg = (int)(g - vox * ct) * V;
    I v = vox;
    const int z = (int)(v % (I)cz); v /= (I)cz;
    const int y = (int)(v % (I)cy); v /= (I)cy;
    const int x = (int)(v % (I)cx);
    const long long b = (long long)(v / (I)cx);
    // candidate fine indices around x * fx / cx
    const int x0 = max(0, (int)((long long)x * fx / cx) - 1), x1 = min(fx - 1, (int)((long long)(x + 1) * fx / cx) + 1);
    const int y0 = max(0, (int)((long long)y * fy / cy) - 1), y1 = min(fy - 1, (int)((long long)(y + 1) * fy / cy) + 1);
    const int z0 = max(0, (int)((long long)z * fz / cz) - 1), z1 = min(fz - 1, (int)((long long)(z + 1) * fz / cz) + 1);
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int a = x0; a <= x1; ++a) {
      if (nearest_src(a, cx, fx) != x) continue;
      for (int bq = y0; bq <= y1; ++bq) {
        if (nearest_src(bq, cy, fy) != y) continue;
        for (int d = z0; d <= z1; ++d) {
          if (nearest_src(d, cz, fz) != z) continue;
          float gv[V];
          vecv<T, V>::ld(dfine + ((((b * fx + a) * fy + bq) * fz + d) * (long long)c + cg), gv);
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += gv[k];
        }
      }
    }
    const long long dst = (long long)vox * c + cg;
    if (accumulate) {
      float old[V];
      vecv<T, V>::ld(dcoarse + dst, old);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += old[k];
    }
    vecv<T, V>::st(dcoarse + dst, acc);
  }
}

// Round 6 fast forms (tensors below 2^31 elements): multiply-shift voxel decomposition (PoolDiv) and the per-axis scale in / out computed once
// on the host (the same correctly rounded fp32 quotient the general kernel forms per lane); the backward of an exact 2x pyramid step (every
// level pair of the three backbones on 160^3-class grids) sums the 8 children 2x + {0,1} directly -- the general kernel finds them by testing
// ~35 candidate indices with nearest_src and six 64-bit divisions per lane.  Same bits: same children, same ascending (x, y, z) order.
__device__ __forceinline__ int nearest_src_s(int dst, int in, float scale) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

template <typename T, int V>
__global__ void __launch_bounds__(256) upsample_add_fwd_fast_kernel(T *__restrict__ fine, const T *__restrict__ coarse, unsigned total, int cx, int cy,
                                                                    int cz, int c, float sx, float sy, float sz, PoolDiv dv) {
  for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
    const unsigned vox = fastdiv(g, dv.ct);
    const int cg = (int)(g - vox * dv.ct.d) * V;
    const unsigned t1 = fastdiv(vox, dv.z), t2 = fastdiv(t1, dv.y), b = fastdiv(t2, dv.x);
    const int z = (int)(vox - t1 * dv.z.d), y = (int)(t1 - t2 * dv.y.d), x = (int)(t2 - b * dv.x.d);
    const long long src = ((((long long)b * cx + nearest_src_s(x, cx, sx)) * cy + nearest_src_s(y, cy, sy)) * cz + nearest_src_s(z, cz, sz)) * (long long)c + cg;
    const long long dst = (long long)vox * c + cg;
    float a[V], bq[V];
    vecv<T, V>::ld(fine + dst, a);
    vecv<T, V>::ld(coarse + src, bq);
#pragma unroll
    for (int k = 0; k < V; ++k) a[k] += bq[k];
    vecv<T, V>::st(fine + dst, a);
  }
}

template <typename T, int V>
__global__ void __launch_bounds__(256) upsample_add_bwd_x2_kernel(const T *__restrict__ dfine, T *__restrict__ dcoarse, unsigned total, int fx, int fy,
                                                                  int fz, int c, int accumulate, PoolDiv dv) {
  for (unsigned g = blockIdx.x * 256u + threadIdx.x; g < total; g += gridDim.x * 256u) {
    const unsigned vox = fastdiv(g, dv.ct);
    const int cg = (int)(g - vox * dv.ct.d) * V;
    const unsigned t1 = fastdiv(vox, dv.z), t2 = fastdiv(t1, dv.y), b = fastdiv(t2, dv.x);
    const int z = (int)(vox - t1 * dv.z.d), y = (int)(t1 - t2 * dv.y.d), x = (int)(t2 - b * dv.x.d);
    float gv[8][V], acc[V];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      vecv<T, V>::ld(dfine + ((((long long)b * fx + 2 * x + (k >> 2)) * fy + 2 * y + ((k >> 1) & 1)) * fz + 2 * z + (k & 1)) * (long long)c + cg, gv[k]);
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int q = 0; q < V; ++q) acc[q] += gv[k][q];
    const long long dst = (long long)vox * c + cg;
    if (accumulate) {
      float old[V];
      vecv<T, V>::ld(dcoarse + dst, old);
#pragma unroll
      for (int q = 0; q < V; ++q) acc[q] += old[q];
    }
    vecv<T, V>::st(dcoarse + dst, acc);
  }
}

extern "C" int nrpn_upsample_add_fwd(void *fine, const void *coarse, int n, int fx, int fy, int fz, int cx, int cy, int cz, int c, int dtype,
                                     nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && fx > 0 && fy > 0 && fz > 0 && cx > 0 && cy > 0 && cz > 0 && c % 4 == 0, "upsample_add_fwd: bad sizes");
  NRPN_REQUIRE(fine && coarse, "upsample_add_fwd: null pointer");
  const bool small = (long long)n * fx * fy * fz * c < (1ll << 31);      // 32-bit index arithmetic; bf16 with C % 8 == 0: 16-byte accesses
#define NRPN_UP_FWD(T_, V_, I_) hipLaunchKernelGGL((upsample_add_fwd_kernel<T_, V_, I_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                   (T_ *)fine, (const T_ *)coarse, n, fx, fy, fz, cx, cy, cz, c)
  if (small && g_pool_fast.load(std::memory_order_relaxed)) {
    const int v = (dtype == NRPN_BF16 && c % 8 == 0) ? 8 : 4;
    const long long total = (long long)n * fx * fy * fz * (c / v);
    const PoolDiv dv{make_fastdiv((unsigned)(c / v)), make_fastdiv((unsigned)fz), make_fastdiv((unsigned)fy), make_fastdiv((unsigned)fx)};
    const float sx = (float)cx / (float)fx, sy = (float)cy / (float)fy, sz = (float)cz / (float)fz;
#define NRPN_UP_FWDF(T_, V_) hipLaunchKernelGGL((upsample_add_fwd_fast_kernel<T_, V_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                (T_ *)fine, (const T_ *)coarse, (unsigned)total, cx, cy, cz, c, sx, sy, sz, dv)
    if (v == 8) NRPN_UP_FWDF(bf16s, 8); else { DISPATCH_T(dtype, NRPN_UP_FWDF(T, 4)); }
#undef NRPN_UP_FWDF
  } else if (dtype == NRPN_BF16 && c % 8 == 0) {
    const long long total = (long long)n * fx * fy * fz * (c / 8);
    if (small) NRPN_UP_FWD(bf16s, 8, unsigned); else NRPN_UP_FWD(bf16s, 8, long long);
  } else {
    const long long total = (long long)n * fx * fy * fz * (c / 4);
    if (small) { DISPATCH_T(dtype, NRPN_UP_FWD(T, 4, unsigned)); } else { DISPATCH_T(dtype, NRPN_UP_FWD(T, 4, long long)); }
  }
#undef NRPN_UP_FWD
  NRPN_LAUNCH_CHECK("upsample_add_fwd");
  return NRPN_OK;
}

extern "C" int nrpn_upsample_add_bwd(const void *dfine, void *dcoarse, int n, int fx, int fy, int fz, int cx, int cy, int cz, int c, int dtype,
                                     int accumulate, nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && fx > 0 && fy > 0 && fz > 0 && cx > 0 && cy > 0 && cz > 0 && c % 4 == 0, "upsample_add_bwd: bad sizes");
  NRPN_REQUIRE(dfine && dcoarse, "upsample_add_bwd: null pointer");
  const bool small = (long long)n * fx * fy * fz * c < (1ll << 31);
#define NRPN_UP_BWD(T_, V_, I_) hipLaunchKernelGGL((upsample_add_bwd_kernel<T_, V_, I_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                   (const T_ *)dfine, (T_ *)dcoarse, n, fx, fy, fz, cx, cy, cz, c, accumulate)
  if (small && fx == 2 * cx && fy == 2 * cy && fz == 2 * cz && g_pool_fast.load(std::memory_order_relaxed)) {
    const int v = (dtype == NRPN_BF16 && c % 8 == 0) ? 8 : 4;
    const long long total = (long long)n * cx * cy * cz * (c / v);
    const PoolDiv dv{make_fastdiv((unsigned)(c / v)), make_fastdiv((unsigned)cz), make_fastdiv((unsigned)cy), make_fastdiv((unsigned)cx)};
#define NRPN_UP_BWDF(T_, V_) hipLaunchKernelGGL((upsample_add_bwd_x2_kernel<T_, V_>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), \
                                                (const T_ *)dfine, (T_ *)dcoarse, (unsigned)total, fx, fy, fz, c, accumulate, dv)
    if (v == 8) NRPN_UP_BWDF(bf16s, 8); else { DISPATCH_T(dtype, NRPN_UP_BWDF(T, 4)); }
#undef NRPN_UP_BWDF
  } else if (dtype == NRPN_BF16 && c % 8 == 0) {
    const long long total = (long long)n * cx * cy * cz * (c / 8);
    if (small) NRPN_UP_BWD(bf16s, 8, unsigned); else NRPN_UP_BWD(bf16s, 8, long long);
  } else {
    const long long total = (long long)n * cx * cy * cz * (c / 4);
    if (small) { DISPATCH_T(dtype, NRPN_UP_BWD(T, 4, unsigned)); } else { DISPATCH_T(dtype, NRPN_UP_BWD(T, 4, long long)); }
  }
#undef NRPN_UP_BWD
  NRPN_LAUNCH_CHECK("upsample_add_bwd");
  return NRPN_OK;
}

// =====================================================================================================================
// strided 1x1x1 convolutions of the ResNet bottlenecks = voxel subsample (every s-th voxel per axis) + plain 1x1x1 GEMM;
// residual join  y = relu(a + b)
// =====================================================================================================================
template <typename T, bool FWD>
__global__ void subsample_kernel(const T *__restrict__ src, T *__restrict__ dst, int n, int gx, int gy, int gz, int ox, int oy, int oz,
                                 int c, int s) {
  // FWD: dst[n,ox,oy,oz,c] = src[n, x*s, y*s, z*s, c];  !FWD: dst[n,gx,gy,gz,c] = (x,y,z all multiples of s) ? src[..] : 0
  const int ct = c / 4;
  const long long total = (long long)n * (FWD ? (long long)ox * oy * oz : (long long)gx * gy * gz) * ct;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    const int cg = (int)(g % ct) * 4;
    long long v = g / ct;
    const int dz = FWD ? oz : gz, dy = FWD ? oy : gy, dx = FWD ? ox : gx;
    const int z = (int)(v % dz); v /= dz;
    const int y = (int)(v % dy); v /= dy;
    const int x = (int)(v % dx);
    const long long b = v / dx;
    if (FWD) {
      vec4<T>::st(dst + (g / ct) * (long long)c + cg, vec4<T>::ld(src + ((((b * gx + x * s) * gy + y * s) * gz + z * s) * (long long)c + cg)));
    } else {
      f4 val = {0, 0, 0, 0};
      if (x % s == 0 && y % s == 0 && z % s == 0) val = vec4<T>::ld(src + ((((b * ox + x / s) * oy + y / s) * oz + z / s) * (long long)c + cg));
      vec4<T>::st(dst + (g / ct) * (long long)c + cg, val);
    }
  }
}

extern "C" int nrpn_subsample3d(const void *src, void *dst, int n, int gx, int gy, int gz, int c, int stride, int backward, int dtype,
                                nrpn_stream_t stream) {
  NRPN_REQUIRE(src && dst && n > 0 && gx > 0 && gy > 0 && gz > 0 && c > 0 && c % 4 == 0 && stride >= 1, "subsample3d: bad args");
  const int ox = (gx - 1) / stride + 1, oy = (gy - 1) / stride + 1, oz = (gz - 1) / stride + 1;
  const long long total = (long long)n * (backward ? (long long)gx * gy * gz : (long long)ox * oy * oz) * (c / 4);
  if (backward) { DISPATCH_T(dtype, hipLaunchKernelGGL((subsample_kernel<T, false>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                                                       (const T *)src, (T *)dst, n, gx, gy, gz, ox, oy, oz, c, stride)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((subsample_kernel<T, true>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                                              (const T *)src, (T *)dst, n, gx, gy, gz, ox, oy, oz, c, stride)); }
  NRPN_LAUNCH_CHECK("subsample3d");
  return NRPN_OK;
}

template <typename T>
__global__ void add_relu_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ y, long long groups, int relu) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (long long)gridDim.x * blockDim.x) {
    const f4 av = vec4<T>::ld(a + g * 4), bv = vec4<T>::ld(b + g * 4);
    f4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = av[k] + bv[k]; if (relu) o[k] = fmaxf(o[k], 0.f); }
    vec4<T>::st(y + g * 4, o);
  }
}

extern "C" int nrpn_add_relu(const void *a, const void *b, void *y, int64_t count, int relu, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(a && b && y && count > 0 && count % 4 == 0, "add_relu: count must be a positive multiple of 4");
  const long long groups = count / 4;
  DISPATCH_T(dtype, hipLaunchKernelGGL(add_relu_kernel<T>, dim3(ew_blocks(groups)), dim3(256), 0, as_stream(stream), (const T *)a,
                                       (const T *)b, (T *)y, groups, relu));
  NRPN_LAUNCH_CHECK("add_relu");
  return NRPN_OK;
}

// =====================================================================================================================
// layout / dtype conversion: [N][C][V] fp32  <->  [N][V][C] T   (32x32 LDS tile transpose)
// =====================================================================================================================
template <typename T, bool TO_CL>
__global__ void transpose_kernel(const void *__restrict__ src, void *__restrict__ dst, int c, long long voxels) {
  __shared__ float tile[32][33];
  const long long b = blockIdx.z;
  const long long v0 = (long long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;  // (32, 8)
  if (TO_CL) {
    const float *s = reinterpret_cast<const float *>(src) + b * c * voxels;
    T *d = reinterpret_cast<T *>(dst) + b * c * voxels;
    for (int i = ty; i < 32; i += 8)
      if (c0 + i < c && v0 + tx < voxels) tile[i][tx] = s[(long long)(c0 + i) * voxels + v0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
      if (v0 + i < voxels && c0 + tx < c) elem<T>::st(d + (v0 + i) * c + c0 + tx, tile[tx][i]);
  } else {
    const T *s = reinterpret_cast<const T *>(src) + b * c * voxels;
    float *d = reinterpret_cast<float *>(dst) + b * c * voxels;
    for (int i = ty; i < 32; i += 8)
      if (v0 + i < voxels && c0 + tx < c) tile[i][tx] = elem<T>::ld(s + (v0 + i) * c + c0 + tx);
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
      if (c0 + i < c && v0 + tx < voxels) d[(long long)(c0 + i) * voxels + v0 + tx] = tile[tx][i];
  }
}

// C == 4 (the rgb-sigma scene itself): one thread per voxel reads the four channel planes (each coalesced across the wave) and writes one
// 8/16-byte channels-last element -- the generic 32x32 LDS transpose wastes 7/8 of its tile on 4 channels (0.9 TB/s).
template <typename T>
__global__ void planes4_to_cl_kernel(const float *__restrict__ src, T *__restrict__ dst, long long voxels, int n) {
  const long long total = voxels * n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / voxels, v = i - b * voxels;
    const float *s = src + b * 4 * voxels + v;
    const f4 o = {s[0], s[voxels], s[2 * voxels], s[3 * voxels]};
    vec4<T>::st(dst + i * 4, o);
  }
}

extern "C" int nrpn_ncdhw_to_ndhwc(const float *src, void *dst, int n, int c, int64_t voxels, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(src && dst && n > 0 && c > 0 && voxels > 0 && n < 65536, "ncdhw_to_ndhwc: bad args");
  if (c == 4) {
    DISPATCH_T(dtype, hipLaunchKernelGGL(planes4_to_cl_kernel<T>, dim3(ew_blocks((long long)voxels * n)), dim3(256), 0, as_stream(stream), src, (T *)dst,
                                         (long long)voxels, n));
    NRPN_LAUNCH_CHECK("ncdhw_to_ndhwc");
    return NRPN_OK;
  }
  dim3 grid((unsigned)cdiv64(voxels, 32), (unsigned)((c + 31) / 32), (unsigned)n);
  DISPATCH_T(dtype, hipLaunchKernelGGL((transpose_kernel<T, true>), grid, dim3(32, 8), 0, as_stream(stream), (const void *)src, dst, c,
                                       (long long)voxels));
  NRPN_LAUNCH_CHECK("ncdhw_to_ndhwc");
  return NRPN_OK;
}

extern "C" int nrpn_ndhwc_to_ncdhw(const void *src, float *dst, int n, int c, int64_t voxels, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(src && dst && n > 0 && c > 0 && voxels > 0 && n < 65536, "ndhwc_to_ncdhw: bad args");
  dim3 grid((unsigned)cdiv64(voxels, 32), (unsigned)((c + 31) / 32), (unsigned)n);
  DISPATCH_T(dtype, hipLaunchKernelGGL((transpose_kernel<T, false>), grid, dim3(32, 8), 0, as_stream(stream), src, (void *)dst, c,
                                       (long long)voxels));
  NRPN_LAUNCH_CHECK("ndhwc_to_ncdhw");
  return NRPN_OK;
}

// Column sums of an f32 [rows][C] matrix, any C (the bias gradient of the bf16x3 mode: sum over voxels of dy).  Deterministic: slabs of rows
// summed in row order per column (coalesced 4-byte loads across adjacent columns), slab partials summed in slab order in fp64.
__global__ void colsum_partial_kernel(const float *__restrict__ x, long long rows, int c, int slab, float *__restrict__ partial) {
  const long long r0 = (long long)blockIdx.x * slab, r1 = min(rows, r0 + slab);
  for (int ch = blockIdx.y * blockDim.x + threadIdx.x; ch < c; ch += gridDim.y * blockDim.x) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long long r = r0;
    for (; r + 3 < r1; r += 4) {
      a0 += x[r * c + ch]; a1 += x[(r + 1) * c + ch]; a2 += x[(r + 2) * c + ch]; a3 += x[(r + 3) * c + ch];
    }
    for (; r < r1; ++r) a0 += x[r * c + ch];
    partial[(long long)blockIdx.x * c + ch] = (a0 + a1) + (a2 + a3);
  }
}
// finish: block = (64 channels, 16 slab lanes), fp64 accumulation in slab order per lane, lanes summed in order (the first version walked all
// slabs with ONE thread per channel: 139 us per call, 3.3 ms of a bf16x3 step)
__global__ void colsum_finish_kernel(const float *__restrict__ partial, int nslabs, int c, float *__restrict__ out, int accumulate) {
  __shared__ double red[16][64];
  const int ch = blockIdx.x * 64 + threadIdx.x;
  double s = 0.0;
  if (ch < c)
    for (int k = threadIdx.y; k < nslabs; k += 16) s += (double)partial[(long long)k * c + ch];
  red[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.y != 0 || ch >= c) return;
  double t = 0.0;
  for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
  out[ch] = accumulate ? out[ch] + (float)t : (float)t;
}
static inline int colsum_slab(long long rows) { const long long s = (rows + 1023) / 1024; return (int)(s < 16 ? 16 : s); }
extern "C" size_t nrpn_column_sum_workspace_bytes(int64_t rows, int c) { return (size_t)(cdiv64(rows, colsum_slab(rows)) * c * 4); }
extern "C" int nrpn_column_sum_f32(const float *x, int64_t rows, int c, float *out, int accumulate, void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(x && out && workspace && rows > 0 && c > 0, "column_sum: bad arguments");
  const int slab = colsum_slab(rows), nslabs = (int)cdiv64(rows, slab);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nslabs, (c + 255) / 256), dim3(256), 0, st, x, (long long)rows, c, slab, (float *)workspace);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((c + 63) / 64), dim3(64, 16), 0, st, (const float *)workspace, nslabs, c, out, accumulate);
  NRPN_LAUNCH_CHECK("column_sum");
  return NRPN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3 ("split bf16") operands of the parity-grade fast mode (round 5): an fp32 value x is carried as hi = bf16(x) and
// lo = bf16(x - hi) (the subtraction is exact in fp32; hi + lo reproduces x to 2^-17 relative), and a product x * w is evaluated as
// hi*whi + hi*wlo + lo*whi on the bf16 MFMA kernels with fp32 accumulation (every bf16 x bf16 product is exact in fp32; what is dropped,
// lo*wlo and the two residuals, is <= 2^-16 relative per product: the size of the fp32 accumulation error of a K = 27 * 256 dot product).
// The conv kernels are NOT changed for it: the K axis is tripled.  Forward / dgrad read an INTERLEAVED operand [rows][3C] whose three
// C-wide segments hold (hi | hi | lo) for activations and (hi | lo | hi) for weights; the weight gradient sums over voxels, so it reads
// PLANES [p][rows][C] stacked on the batch axis -- x as (hi ; lo ; hi), dy as (hi ; hi ; lo) -- which the wgrad kernels sum like scenes.
// One pass over the fp32 source writes either or both forms.  Segment / plane s holds lo when bit s of the pattern is set, else hi.
// 8 channels per lane: two 16-byte loads, 16-byte stores (C % 8 == 0).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void split_bf16x3_kernel(const float *__restrict__ src, long long rows, int c, bf16s *__restrict__ inter, int ipat,
                                    bf16s *__restrict__ planes, int nplanes, int ppat) {
  typedef __attribute__((ext_vector_type(8))) unsigned short u8v;
  const int groups = c >> 3;
  const long long total = rows * groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / groups;
    const int g = (int)(i - r * groups);
    const f4 a = *reinterpret_cast<const f4 *>(src + r * c + g * 8), b = *reinterpret_cast<const f4 *>(src + r * c + g * 8 + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    u8v hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned short h = f32_to_bf16_bits(v[e]);
      hi[e] = h;
      lo[e] = f32_to_bf16_bits(v[e] - bf16_bits_to_f32(h));
    }
    if (inter) {
      bf16s *d = inter + r * 3 * c + g * 8;
#pragma unroll
      for (int sgm = 0; sgm < 3; ++sgm) *reinterpret_cast<u8v *>(d + sgm * c) = ((ipat >> sgm) & 1) ? lo : hi;
    }
    if (planes) {
      bf16s *d = planes + r * c + g * 8;
      for (int pl = 0; pl < nplanes; ++pl) *reinterpret_cast<u8v *>(d + (long long)pl * rows * c) = ((ppat >> pl) & 1) ? lo : hi;
    }
  }
}

extern "C" int nrpn_split_bf16x3(const float *src, int64_t rows, int c, void *interleaved, int ipattern, void *planes, int nplanes, int ppattern,
                                 nrpn_stream_t stream) {
  NRPN_REQUIRE(src && rows > 0 && c > 0 && (c & 7) == 0, "split_bf16x3: rows > 0 and C a multiple of 8 (got C = %d)", c);
  NRPN_REQUIRE(interleaved || planes, "split_bf16x3: nothing to write");
  NRPN_REQUIRE(!planes || (nplanes >= 1 && nplanes <= 3), "split_bf16x3: 1..3 planes (got %d)", nplanes);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(ew_blocks((long long)rows * (c >> 3))), dim3(256), 0, as_stream(stream), src, (long long)rows, c,
                     reinterpret_cast<bf16s *>(interleaved), ipattern, reinterpret_cast<bf16s *>(planes), nplanes, ppattern);
  NRPN_LAUNCH_CHECK("split_bf16x3");
  return NRPN_OK;
}

template <typename S, typename D>
__global__ void cast_kernel(const S *__restrict__ s, D *__restrict__ d, long long count) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
    elem<D>::st(d + i, elem<S>::ld(s + i));
}

extern "C" int nrpn_cast(const void *src, void *dst, int64_t count, int src_dtype, int dst_dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(src && dst && count > 0, "cast: bad args");
  hipStream_t st = as_stream(stream);
  const dim3 grid(ew_blocks(count));
  if (src_dtype == NRPN_F32 && dst_dtype == NRPN_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16s>), grid, dim3(256), 0, st, (const float *)src, (bf16s *)dst, (long long)count);
  else if (src_dtype == NRPN_BF16 && dst_dtype == NRPN_F32)
    hipLaunchKernelGGL((cast_kernel<bf16s, float>), grid, dim3(256), 0, st, (const bf16s *)src, (float *)dst, (long long)count);
  else if (src_dtype == NRPN_F32 && dst_dtype == NRPN_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), grid, dim3(256), 0, st, (const float *)src, (float *)dst, (long long)count);
  else
    hipLaunchKernelGGL((cast_kernel<bf16s, bf16s>), grid, dim3(256), 0, st, (const bf16s *)src, (bf16s *)dst, (long long)count);
  NRPN_LAUNCH_CHECK("cast");
  return NRPN_OK;
}

// =====================================================================================================================
// optimiser on the flat fp32 arena
// =====================================================================================================================
// gradient exchange helper: the reduction step of the bf16 all-to-all mode (engine.FlatTrainer, exchange = "a2a_bf16")
// =====================================================================================================================
// out[i] = bf16( sum over ranks r of (r == rank ? local_f32[i] : float(recv[r][i])) ), ranks in ascending order, fp32 accumulation: the
// local contribution never goes through bf16, the result is rounded once.  One pass (world x 2 B + 4 B read, 2 B written per element)
// instead of cast + slice-assign + sum + cast.
__global__ void __launch_bounds__(256) a2a_reduce_kernel(const bf16s *__restrict__ recv, const float *__restrict__ local, int rank, int world,
                                                         long long chunk, bf16s *__restrict__ out) {
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < chunk; i += (long long)gridDim.x * blockDim.x * 4) {
    if (i + 4 <= chunk) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < world; ++r) {
        const f4 v = (r == rank) ? *reinterpret_cast<const f4 *>(local + i) : vec4<bf16s>::ld(recv + (long long)r * chunk + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += v[k];
      }
      vec4<bf16s>::st(out + i, acc);
    } else {
      for (long long e = i; e < chunk; ++e) {
        float acc = 0.f;
        for (int r = 0; r < world; ++r) acc += (r == rank) ? local[e] : bf16_bits_to_f32(recv[(long long)r * chunk + e]);
        out[e] = f32_to_bf16_bits(acc);
      }
    }
  }
}

extern "C" int nrpn_a2a_reduce_bf16(const void *recv, const float *local, int rank, int world, int64_t chunk, void *out, nrpn_stream_t stream) {
  NRPN_REQUIRE(recv && local && out && world >= 1 && rank >= 0 && rank < world && chunk > 0 && chunk % 4 == 0,
               "a2a_reduce_bf16: bad arguments (chunk must be a multiple of 4 elements)");
  const long long groups = chunk / 4;
  long long blocks = (groups + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(a2a_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), (const bf16s *)recv, local, rank, world, (long long)chunk,
                     (bf16s *)out);
  NRPN_LAUNCH_CHECK("a2a_reduce_bf16");
  return NRPN_OK;
}

// =====================================================================================================================
// Deterministic sum of squares: one fixed-order partial per block, added in index order by sumsq_finish_kernel.
// buf: [0] result, [1] unused (was a ticket counter), [2 .. 2 + kSumsqBlocks) partials.
constexpr int kSumsqBlocks = 2048;       // capacity of the partials buffer; the launched grid is g_sumsq_grid (<= this)
static int g_sumsq_grid = 1024, g_sumsq_form = 0;   // tools switch nrpn_set_sumsq_form: A/B only, defaults are the measured best

// FORM 0: each block owns one contiguous range (four 16-byte loads in flight per lane).  FORM 1: same ranges, eight loads in flight.
// FORM 2: grid-stride -- at any instant the whole grid reads one moving window (what adamw_kernel does), eight loads in flight.
template <int FORM>
__global__ void __launch_bounds__(256) sumsq_kernel(const float *__restrict__ g, long long count, float scale, float *__restrict__ buf) {
  __shared__ float sh[256];
  const long long n4 = count / 4;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const f4 *g4 = reinterpret_cast<const f4 *>(g);
  auto acc4 = [&](const f4 &a, const f4 &b, const f4 &c, const f4 &d) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float ta = a[k] * scale, tb = b[k] * scale, tc = c[k] * scale, td = d[k] * scale;
      s0 += ta * ta; s1 += tb * tb; s2 += tc * tc; s3 += td * td;
    }
  };
  if constexpr (FORM == 2) {
    const long long S = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * S < n4; i += 8 * S) {
      const f4 a = g4[i], b = g4[i + S], c = g4[i + 2 * S], d = g4[i + 3 * S];
      const f4 a2 = g4[i + 4 * S], b2 = g4[i + 5 * S], c2 = g4[i + 6 * S], d2 = g4[i + 7 * S];
      acc4(a, b, c, d);
      acc4(a2, b2, c2, d2);
    }
    for (; i < n4; i += S) {
      const f4 a = g4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float ta = a[k] * scale; s0 += ta * ta; }
    }
  } else {
    const long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long long b0 = (long long)blockIdx.x * per, b1 = min(n4, b0 + per);
    long long i = b0 + threadIdx.x;
    if constexpr (FORM == 1) {
      for (; i + 1792 < b1; i += 2048) {
        const f4 a = g4[i], b = g4[i + 256], c = g4[i + 512], d = g4[i + 768];
        const f4 a2 = g4[i + 1024], b2 = g4[i + 1280], c2 = g4[i + 1536], d2 = g4[i + 1792];
        acc4(a, b, c, d);
        acc4(a2, b2, c2, d2);
      }
    }
    for (; i + 768 < b1; i += 1024) {       // four independent 16-byte loads in flight per lane
      const f4 a = g4[i], b = g4[i + 256], c = g4[i + 512], d = g4[i + 768];
      acc4(a, b, c, d);
    }
    for (; i < b1; i += 256) {
      const f4 a = g4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float ta = a[k] * scale; s0 += ta * ta; }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(count - n4 * 4)) { const float t = g[n4 * 4 + threadIdx.x] * scale; s1 += t * t; }
  sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) buf[2 + blockIdx.x] = sh[0];
}

// Adds the per-block partials in index order (one block, fixed tree).  Round 5: this used to be the tail of sumsq_kernel behind a ticket counter
// (threadfence + atomicAdd per block); measured, the ticket cost ~50 ns PER BLOCK, serialised (61 / 86 / 138 us at 512 / 1024 / 2048 blocks whatever
// the read pattern: a device-scope release on gfx950 writes back the XCD's L2) -- more than the 37 us the 299 MB read takes.  A second launch is ~3 us.
__global__ void __launch_bounds__(256) sumsq_finish_kernel(float *__restrict__ buf, int parts) {
  __shared__ float sh[256];
  float t = 0.f;
  for (int k = threadIdx.x; k < parts; k += 256) t += buf[2 + k];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) buf[0] = sh[0];
}

extern "C" int nrpn_set_sumsq_form(int form, int grid) {
  NRPN_REQUIRE(form >= 0 && form <= 2 && grid >= 64 && grid <= kSumsqBlocks, "set_sumsq_form: form in 0..2, grid in 64..2048");
  g_sumsq_form = form;
  g_sumsq_grid = grid;
  return NRPN_OK;
}

extern "C" int nrpn_grad_sumsq_floats(void) { return 2 + kSumsqBlocks; }

extern "C" int nrpn_grad_sumsq(const float *grad, int64_t count, float grad_scale, float *sumsq, nrpn_stream_t stream) {
  NRPN_REQUIRE(grad && sumsq && count > 0, "grad_sumsq: bad args");
  NRPN_REQUIRE((reinterpret_cast<uintptr_t>(grad) & 15) == 0, "grad_sumsq: the gradient arena must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const dim3 grid(g_sumsq_grid);
  if (g_sumsq_form == 2) hipLaunchKernelGGL(sumsq_kernel<2>, grid, dim3(256), 0, st, grad, (long long)count, grad_scale, sumsq);
  else if (g_sumsq_form == 1) hipLaunchKernelGGL(sumsq_kernel<1>, grid, dim3(256), 0, st, grad, (long long)count, grad_scale, sumsq);
  else hipLaunchKernelGGL(sumsq_kernel<0>, grid, dim3(256), 0, st, grad, (long long)count, grad_scale, sumsq);
  hipLaunchKernelGGL(sumsq_finish_kernel, dim3(1), dim3(256), 0, st, sumsq, g_sumsq_grid);
  NRPN_LAUNCH_CHECK("grad_sumsq");
  return NRPN_OK;
}

// 16 bytes per lane per stream (p, g, m, v read; p, m, v [, g = 0] [, bf16 shadow] written); ZERO: the consumed gradient is cleared in
// the same pass (the next backward accumulates into a clean arena without a separate 299 MB fill)
template <bool ZERO>
__global__ void adamw_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v, long long count,
                             const float *__restrict__ sumsq, float grad_scale, float max_norm, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, bf16s *__restrict__ shadow) {
  float coef = grad_scale;
  if (sumsq && max_norm > 0.f) {   // sumsq[0] = the ordered total left by nrpn_grad_sumsq
    const float norm = sqrtf(*sumsq);
    coef = grad_scale * fminf(1.0f, max_norm / (norm + 1e-6f));
  }
  const float decay = 1.0f - lr * wd, step = lr / bc1;
  auto upd = [&](float gi, float &pi, float &mi, float &vi) {
    gi *= coef;
    pi *= decay;
    mi = b1 * mi + (1.0f - b1) * gi;
    vi = b2 * vi + (1.0f - b2) * gi * gi;
    pi -= step * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  };
  typedef __attribute__((ext_vector_type(4))) float f4;
  const long long quads = count >> 2;
  const long long S = (long long)gridDim.x * blockDim.x;
  auto quad = [&](long long q, f4 pv, f4 mv, f4 vv, const f4 gv) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = pv[e], me = mv[e], ve = vv[e];
      upd(gv[e], pe, me, ve);
      pv[e] = pe; mv[e] = me; vv[e] = ve;
    }
    reinterpret_cast<f4 *>(p)[q] = pv; reinterpret_cast<f4 *>(m)[q] = mv; reinterpret_cast<f4 *>(v)[q] = vv;
    if (ZERO) reinterpret_cast<f4 *>(g)[q] = f4{0.f, 0.f, 0.f, 0.f};
    if (shadow) {      // bf16 copy of the updated master weights, same element order
      typedef __attribute__((ext_vector_type(4))) unsigned short u4s;
      u4s h = {f32_to_bf16_bits(pv[0]), f32_to_bf16_bits(pv[1]), f32_to_bf16_bits(pv[2]), f32_to_bf16_bits(pv[3])};
      reinterpret_cast<u4s *>(shadow)[q] = h;
    }
  };
  long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // two quads per iteration: eight 16-byte loads in flight per lane before the first store (round 6: 0.66 -> see bench hbm_stages)
  for (; q + S < quads; q += 2 * S) {
    const f4 p0 = reinterpret_cast<f4 *>(p)[q], m0 = reinterpret_cast<f4 *>(m)[q], v0 = reinterpret_cast<f4 *>(v)[q];
    const f4 g0 = reinterpret_cast<const f4 *>(g)[q];
    const f4 p1 = reinterpret_cast<f4 *>(p)[q + S], m1 = reinterpret_cast<f4 *>(m)[q + S], v1 = reinterpret_cast<f4 *>(v)[q + S];
    const f4 g1 = reinterpret_cast<const f4 *>(g)[q + S];
    quad(q, p0, m0, v0, g0);
    quad(q + S, p1, m1, v1, g1);
  }
  for (; q < quads; q += S) {
    quad(q, reinterpret_cast<f4 *>(p)[q], reinterpret_cast<f4 *>(m)[q], reinterpret_cast<f4 *>(v)[q], reinterpret_cast<const f4 *>(g)[q]);
  }
  for (long long i = (quads << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(g[i], pi, mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (ZERO) g[i] = 0.f;
    if (shadow) shadow[i] = f32_to_bf16_bits(pi);
  }
}

static int adamw_launch(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t count, const float *sumsq, float grad_scale,
                        float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step, void *shadow_bf16, bool zero,
                        nrpn_stream_t stream) {
  NRPN_REQUIRE(param && grad && exp_avg && exp_avg_sq && count > 0 && step >= 1, "adamw_step: bad args");
  NRPN_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 && (!shadow_bf16 || (uintptr_t)shadow_bf16 % 8 == 0),
               "adamw_step: arenas must be 16-byte aligned (bf16 shadow: 8)");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  const dim3 grid(ew_blocks((count + 3) / 4));
  if (zero)
    hipLaunchKernelGGL(adamw_kernel<true>, grid, dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, (long long)count, sumsq, grad_scale,
                       max_norm, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), reinterpret_cast<bf16s *>(shadow_bf16));
  else
    hipLaunchKernelGGL(adamw_kernel<false>, grid, dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, (long long)count, sumsq, grad_scale,
                       max_norm, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), reinterpret_cast<bf16s *>(shadow_bf16));
  NRPN_LAUNCH_CHECK("adamw_step");
  return NRPN_OK;
}

extern "C" int nrpn_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t count, const float *sumsq,
                               float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               void *shadow_bf16, nrpn_stream_t stream) {
  return adamw_launch(param, const_cast<float *>(grad), exp_avg, exp_avg_sq, count, sumsq, grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay, step,
                      shadow_bf16, false, stream);
}

// nrpn_adamw_step that also clears the gradient it has just consumed (grad[i] = 0): one pass instead of the step + a fill of the arena
extern "C" int nrpn_adamw_step_zero_grad(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t count, const float *sumsq,
                                         float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                         int step, void *shadow_bf16, nrpn_stream_t stream) {
  return adamw_launch(param, grad, exp_avg, exp_avg_sq, count, sumsq, grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay, step, shadow_bf16, true,
                      stream);
}

// =====================================================================================================================
// GEMM-layout master weights.  A trainer that keeps conv weights in the forward GEMM layout [taps][Cout][Cin] inside its flat
// arena needs (a) the dgrad operand [taps reversed][Cin][Cout] of every weight -- ONE batched launch per optimiser step over a
// device-side job table instead of one pack launch per layer -- and (b) the sum of a wgrad's voxel-slice partials, which are
// already in that layout, added straight into the gradient arena (contiguous; no layout shuffle).
// table: int64 [nweights][4] = (element offset in the arena, taps, cout, cin); tile_prefix: int32 [nweights + 1] = first 64x64-tile
// job of each weight (jobs of a weight: taps * ceil(cout/64) * ceil(cin/64)).
// =====================================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) transpose_weights_kernel(const float *__restrict__ master, T *__restrict__ dst, const long long *__restrict__ table,
                                                                const int *__restrict__ tile_prefix, int nweights) {
  __shared__ float tile[64][65];
  const int job = blockIdx.x;
  int lo = 0, hi = nweights - 1;                 // last weight whose first job <= this job
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_prefix[mid] <= job) lo = mid; else hi = mid - 1;
  }
  const long long off = table[lo * 4];
  const int taps = (int)table[lo * 4 + 1], cout = (int)table[lo * 4 + 2], cin = (int)table[lo * 4 + 3];
  int j = job - tile_prefix[lo];
  const int tn = (cin + 63) / 64, tm = (cout + 63) / 64;
  const int c0 = (j % tn) * 64; j /= tn;
  const int o0 = (j % tm) * 64; j /= tm;
  const int tap = j;
  const float *src = master + off + (long long)tap * cout * cin;
  T *out = dst + off + (long long)(taps - 1 - tap) * cin * cout;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4)
    tile[r][tx] = (o0 + r < cout && c0 + tx < cin) ? src[(long long)(o0 + r) * cin + c0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 64; r += 4)
    if (c0 + r < cin && o0 + tx < cout) elem<T>::st(out + (long long)(c0 + r) * cout + o0 + tx, tile[tx][r]);
}

extern "C" int nrpn_transpose_weights(const float *master, void *dst, const int64_t *table, const int32_t *tile_prefix, int nweights,
                                      int total_tiles, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(master && dst && table && tile_prefix && nweights > 0 && total_tiles > 0, "transpose_weights: bad args");
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "transpose_weights: bad dtype %d", dtype);
  hipStream_t st = as_stream(stream);
  if (dtype == NRPN_F32)
    hipLaunchKernelGGL(transpose_weights_kernel<float>, dim3(total_tiles), dim3(256), 0, st, master, (float *)dst, (const long long *)table, tile_prefix, nweights);
  else
    hipLaunchKernelGGL(transpose_weights_kernel<bf16s>, dim3(total_tiles), dim3(256), 0, st, master, (bf16s *)dst, (const long long *)table, tile_prefix, nweights);
  NRPN_LAUNCH_CHECK("transpose_weights");
  return NRPN_OK;
}

// dst[i] (+)= sum_s part[s][i]   (ordered: deterministic).  The last block optionally finishes the bias gradient of the same wgrad
// (per-slice column sums left in its workspace), so weight + bias gradients of a layer cost one launch.
__global__ void __launch_bounds__(256) reduce_slices_kernel(const float *__restrict__ part, int slices, long long count4, long long stride4,
                                                            float *__restrict__ dst, int accumulate, const float *__restrict__ bias_part,
                                                            int wrows, int cout, float *__restrict__ gbias, int accumulate_bias) {
  if (bias_part && blockIdx.x == gridDim.x - 1) {
    for (int c = threadIdx.x; c < cout; c += 256) {
      float s = 0.f;
      for (int k = 0; k < slices; ++k) s += bias_part[(long long)k * wrows + c];
      gbias[c] = accumulate_bias ? gbias[c] + s : s;
    }
    return;
  }
  const f4 *p4 = reinterpret_cast<const f4 *>(part);
  f4 *d4 = reinterpret_cast<f4 *>(dst);
  const long long nb = gridDim.x - (bias_part ? 1 : 0);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count4; i += nb * blockDim.x) {
    f4 a = p4[i];
    for (int s = 1; s < slices; ++s) { const f4 b = p4[(long long)s * stride4 + i]; a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3]; }
    if (accumulate) { const f4 o = d4[i]; a[0] += o[0]; a[1] += o[1]; a[2] += o[2]; a[3] += o[3]; }
    d4[i] = a;
  }
}

extern "C" int nrpn_reduce_slices(const float *partials, int slices, int64_t count, float *dst, int accumulate, const float *bias_partials,
                                  int wrows, int cout, float *gbias, int accumulate_bias, nrpn_stream_t stream) {
  NRPN_REQUIRE(partials && dst && slices > 0 && count > 0 && count % 4 == 0, "reduce_slices: count must be a positive multiple of 4");
  NRPN_REQUIRE(((reinterpret_cast<uintptr_t>(partials) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, "reduce_slices: 16-byte alignment required");
  NRPN_REQUIRE(!bias_partials || (gbias && cout > 0 && wrows >= cout), "reduce_slices: bad bias arguments");
  hipLaunchKernelGGL(reduce_slices_kernel, dim3(ew_blocks(count / 4) + (bias_partials ? 1 : 0)), dim3(256), 0, as_stream(stream), partials, slices,
                     (long long)(count / 4), (long long)(count / 4), dst, accumulate, bias_partials, wrows, cout, gbias, accumulate_bias);
  NRPN_LAUNCH_CHECK("reduce_slices");
  return NRPN_OK;
}

// =====================================================================================================================
// scene ingest (reference datasets.py:39-63, 165-167, 227-231): the on-disk (W,L,H,4) rgb-sigma array is already the
// channels-last activation layout -- one pass applies uint8 -> /255, density_to_alpha on channel 3 and the cast to the
// compute dtype, instead of numpy alpha + host transpose + float conversion + a device transpose back.
// =====================================================================================================================
template <typename S, typename T>
__global__ void ingest_kernel(const S *__restrict__ src, T *__restrict__ dst, long long voxels, int mode) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (long long)gridDim.x * blockDim.x) {
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = sizeof(S) == 1 ? (float)src[v * 4 + k] * (1.0f / 255.0f) : (float)src[v * 4 + k];
    if (mode == 1) c[3] = fminf(fmaxf(1.0f - expf(-expf(c[3]) / 100.0f), 0.f), 1.f);                 // density_to_alpha
    else if (mode == 2) c[3] = fminf(fmaxf(1.0f - expf(-fmaxf(c[3], 0.f) / 100.0f), 0.f), 1.f);     // ScanNet variant (relu)
    f4 o = {c[0], c[1], c[2], c[3]};
    vec4<T>::st(dst + v * 4, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Ingest + augmentation in one pass (reference datasets.py:109-163, 291-329): the 90-degree rotation and the axis flips are index
// remaps, rotate_and_scale_scene is a trilinear resample (F.grid_sample, align_corners=True, zero padding) of the scene AFTER
// density_to_alpha -- so every tap of the interpolation goes through the same load + alpha conversion as the plain ingest.
// Output voxel (i,j,k) of the augmented grid [OW,OL,OH]:  rotate-scale (optional) -> flips -> rotation -> source voxel (w,l,h).
// ---------------------------------------------------------------------------------------------------------------------
struct AugArgs {
  int W, L, H;          // source grid (on-disk order)
  int OW, OL, OH;       // augmented grid
  int rot, z_up;        // rot: 90-degree rotation (z_up: transpose(x,y) + flip x; else transpose(x,z) + flip z)
  int flip0, flip1;     // flips of axis 0 and of axis 1 (z_up) / 2 (not z_up), applied after the rotation
  int rs;               // rotate_and_scale_scene active
  float m[9];           // xform = R(angle) * scale, row-major (datasets.py:294-298)
};

template <typename S>
__device__ __forceinline__ f4 aug_fetch(const S *__restrict__ src, const AugArgs &a, int i, int j, int k, int mode) {
  // undo the flips (they act on the rotated grid [OW,OL,OH]) ...
  if (a.flip0) i = a.OW - 1 - i;
  if (a.flip1) { if (a.z_up) j = a.OL - 1 - j; else k = a.OH - 1 - k; }
  // ... and the rotation: z_up   out[i,j,k] = in[j, L-1-i, k];   else   out[i,j,k] = in[W-1-k, j, i]
  int w = i, l = j, h = k;
  if (a.rot) {
    if (a.z_up) { w = j; l = a.L - 1 - i; h = k; }
    else { w = a.W - 1 - k; l = j; h = i; }
  }
  const long long v = ((long long)w * a.L + l) * a.H + h;
  float c[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) c[q] = sizeof(S) == 1 ? (float)src[v * 4 + q] * (1.0f / 255.0f) : (float)src[v * 4 + q];
  if (mode == 1) c[3] = fminf(fmaxf(1.0f - expf(-expf(c[3]) / 100.0f), 0.f), 1.f);
  else if (mode == 2) c[3] = fminf(fmaxf(1.0f - expf(-fmaxf(c[3], 0.f) / 100.0f), 0.f), 1.f);
  f4 o = {c[0], c[1], c[2], c[3]};
  return o;
}

template <typename S, typename T>
__global__ void ingest_augment_kernel(const S *__restrict__ src, T *__restrict__ dst, AugArgs a, int mode) {
  const long long total = (long long)a.OW * a.OL * a.OH;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(v % a.OH);
    const int j = (int)((v / a.OH) % a.OL);
    const int i = (int)(v / ((long long)a.OH * a.OL));
    f4 o;
    if (!a.rs) {
      o = aug_fetch<S>(src, a, i, j, k, mode);
    } else {
      // grid point of the reference: linspace(-1, 1, n)[idx] * n / 2 per axis, times xform^T, normalised by n/2 again
      const float x = (a.OW > 1 ? -1.0f + (float)i * (2.0f / (float)(a.OW - 1)) : -1.0f) * (float)a.OW / 2;
      const float y = (a.OL > 1 ? -1.0f + (float)j * (2.0f / (float)(a.OL - 1)) : -1.0f) * (float)a.OL / 2;
      const float z = (a.OH > 1 ? -1.0f + (float)k * (2.0f / (float)(a.OH - 1)) : -1.0f) * (float)a.OH / 2;
      const float px = x * a.m[0] + y * a.m[1] + z * a.m[2];
      const float py = x * a.m[3] + y * a.m[4] + z * a.m[5];
      const float pz = x * a.m[6] + y * a.m[7] + z * a.m[8];
      // grid_sample, align_corners=True: index = (g + 1) / 2 * (size - 1)
      const float fx = ((px / ((float)a.OW / 2)) + 1.f) / 2.f * (float)(a.OW - 1);
      const float fy = ((py / ((float)a.OL / 2)) + 1.f) / 2.f * (float)(a.OL - 1);
      const float fz = ((pz / ((float)a.OH / 2)) + 1.f) / 2.f * (float)(a.OH - 1);
      const float x0f = floorf(fx), y0f = floorf(fy), z0f = floorf(fz);
      const int x0 = (int)x0f, y0 = (int)y0f, z0 = (int)z0f;
      const float wx1 = fx - x0f, wy1 = fy - y0f, wz1 = fz - z0f;
      const float wx0 = (x0f + 1.f) - fx, wy0 = (y0f + 1.f) - fy, wz0 = (z0f + 1.f) - fz;
      o = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dz = 0; dz < 2; ++dz) {
            const int xi = x0 + dx, yi = y0 + dy, zi = z0 + dz;
            if ((unsigned)xi < (unsigned)a.OW && (unsigned)yi < (unsigned)a.OL && (unsigned)zi < (unsigned)a.OH) {
              const float wgt = (dx ? wx1 : wx0) * (dy ? wy1 : wy0) * (dz ? wz1 : wz0);
              const f4 t = aug_fetch<S>(src, a, xi, yi, zi, mode);
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] += t[q] * wgt;
            }
          }
    }
    vec4<T>::st(dst + v * 4, o);
  }
}

extern "C" int nrpn_ingest_augment(const void *src, int src_is_u8, void *dst, int w, int l, int h, int alpha_mode, int dtype, int rot90,
                                   int z_up, int flip0, int flip1, const float *h_xform, nrpn_stream_t stream) {
  NRPN_REQUIRE(src && dst && w > 0 && l > 0 && h > 0 && alpha_mode >= 0 && alpha_mode <= 2, "ingest_augment: bad args");
  NRPN_REQUIRE(!(src_is_u8 && alpha_mode), "ingest_augment: density_to_alpha on uint8 grids is only available on the host path");
  AugArgs a{};
  a.W = w; a.L = l; a.H = h;
  a.rot = rot90 ? 1 : 0; a.z_up = z_up ? 1 : 0; a.flip0 = flip0 ? 1 : 0; a.flip1 = flip1 ? 1 : 0;
  if (a.rot) { if (a.z_up) { a.OW = l; a.OL = w; a.OH = h; } else { a.OW = h; a.OL = l; a.OH = w; } }
  else { a.OW = w; a.OL = l; a.OH = h; }
  a.rs = h_xform ? 1 : 0;
  if (h_xform) for (int q = 0; q < 9; ++q) a.m[q] = h_xform[q];
  const long long total = (long long)w * l * h;
  const int blocks = (int)min((long long)8192, (total + 255) / 256);
  hipStream_t st = as_stream(stream);
  if (src_is_u8) { DISPATCH_T(dtype, hipLaunchKernelGGL((ingest_augment_kernel<unsigned char, T>), dim3(blocks), dim3(256), 0, st, (const unsigned char *)src, (T *)dst, a, alpha_mode)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((ingest_augment_kernel<float, T>), dim3(blocks), dim3(256), 0, st, (const float *)src, (T *)dst, a, alpha_mode)); }
  NRPN_LAUNCH_CHECK("ingest_augment");
  return NRPN_OK;
}

extern "C" int nrpn_ingest_rgbsigma(const void *src, int src_is_u8, void *dst, int64_t voxels, int alpha_mode, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(src && dst && voxels > 0 && alpha_mode >= 0 && alpha_mode <= 2, "ingest_rgbsigma: bad args");
  NRPN_REQUIRE(!(src_is_u8 && alpha_mode), "ingest_rgbsigma: density_to_alpha on uint8 grids follows numpy's float16/uint8 casts in the "
                                           "reference and is only available on the host path");
  const int blocks = (int)min((long long)8192, (long long)((voxels + 255) / 256));
  hipStream_t st = as_stream(stream);
  if (src_is_u8) { DISPATCH_T(dtype, hipLaunchKernelGGL((ingest_kernel<unsigned char, T>), dim3(blocks), dim3(256), 0, st, (const unsigned char *)src, (T *)dst, (long long)voxels, alpha_mode)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((ingest_kernel<float, T>), dim3(blocks), dim3(256), 0, st, (const float *)src, (T *)dst, (long long)voxels, alpha_mode)); }
  NRPN_LAUNCH_CHECK("ingest_rgbsigma");
  return NRPN_OK;
}

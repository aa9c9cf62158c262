// Swin-3D building blocks for gfx950 (reference nerf_rpn/model/feature_extractor.py:382-789): patch embedding gather,
// LayerNorm, exact GELU, residual join with per-sample stochastic-depth scale, patch-merging gather, and shifted-window
// multi-head attention (4x4x4 windows, head_dim 32) forward + backward.
//
// Tokens stay in the channels-last activation layout [N][X][Y][Z][C]; every token-wise Linear (qkv, proj, MLP, reduction)
// is the 1x1x1 MFMA GEMM of conv3d.hip.  Window partition, cyclic shift, zero padding to a window multiple and their
// inverses are pure index arithmetic inside the attention kernels -- nothing is rolled, padded or permuted in memory.
// The attention matmuls are 64x32x64 per (window, head) ("attention GEMMs are tiny"): one wavefront per
// (window, head), one lane per query token, K/V staged in LDS and broadcast-read, softmax in registers.
#include "common.h"

typedef unsigned short bf16s;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned short us4;

#define DISPATCH_T(dtype, ...)                                 \
  if ((dtype) == NRPN_F32) { typedef float T; __VA_ARGS__; }   \
  else { typedef bf16s T; __VA_ARGS__; }

static inline int ew_blocks(long long work) { long long b = (work + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

// =====================================================================================================================
// patch embedding gather: [N,X,Y,Z,4] -> [N,X/p,Y/p,Z/p, 4*p^3] with inner order (c, dx, dy, dz) = the flattened
// Conv3d(4, E, k=p, s=p) weight, so the conv becomes a 1x1x1 GEMM on the weight viewed as [E, 4 p^3]
// =====================================================================================================================
template <typename T>
__global__ void patchify_kernel(const T *__restrict__ x, T *__restrict__ y, int n, int gx, int gy, int gz, int ox, int oy, int oz, int p) {
  const int per = 4 * p * p * p;
  const long long total = (long long)n * ox * oy * oz * per;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int k = (int)(i % per);
    long long v = i / per;
    const int dz = k % p; k /= p;
    const int dy = k % p; k /= p;
    const int dx = k % p;
    const int c = k / p;
    const int z = (int)(v % oz); v /= oz;
    const int yy = (int)(v % oy); v /= oy;
    const int xx = (int)(v % ox);
    const long long b = v / ox;
    y[i] = x[(((b * gx + xx * p + dx) * gy + yy * p + dy) * gz + z * p + dz) * 4 + c];
  }
}

extern "C" int nrpn_patchify(const void *x, void *y, int n, int gx, int gy, int gz, int patch, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(x && y && n > 0 && patch >= 1 && gx >= patch && gy >= patch && gz >= patch, "patchify: bad args");
  const int ox = gx / patch, oy = gy / patch, oz = gz / patch;
  const long long total = (long long)n * ox * oy * oz * 4 * patch * patch * patch;
  DISPATCH_T(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), (const T *)x, (T *)y, n,
                                       gx, gy, gz, ox, oy, oz, patch));
  NRPN_LAUNCH_CHECK("patchify");
  return NRPN_OK;
}

// =====================================================================================================================
// LayerNorm over the channel dimension: one wavefront per token row
// =====================================================================================================================
__device__ __forceinline__ float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <typename T>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const T *__restrict__ x, T *__restrict__ y, const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *__restrict__ mean,
                                                            float *__restrict__ rstd, long long rows, int c, float eps) {
  const int lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
    const T *xr = x + r * c;
    float s = 0.f;
    for (int k = lane; k < c; k += 64) s += elem<T>::ld(xr + k);
    const float m = wave_sum(s) / (float)c;
    float q = 0.f;
    for (int k = lane; k < c; k += 64) { const float d = elem<T>::ld(xr + k) - m; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)c + eps);
    for (int k = lane; k < c; k += 64) elem<T>::st(y + r * c + k, (elem<T>::ld(xr + k) - m) * rs * gamma[k] + beta[k]);
    if (lane == 0) { mean[r] = m; rstd[r] = rs; }
  }
}

// dx per row; dgamma / dbeta: per-lane partial sums over the block's rows, reduced over the block's 4 waves through LDS and
// written as one partial row per block (no atomics: ~1000 same-address fp32 atomics per channel made this kernel 3x slower,
// and the result is now run-to-run deterministic); layernorm_param_kernel sums the block partials.
template <typename T, int MAXK>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ dx,
                                                            const float *__restrict__ gamma, const float *__restrict__ mean,
                                                            const float *__restrict__ rstd, float *__restrict__ partial, long long rows,
                                                            int c) {
  __shared__ float red[3][2][64 * (MAXK > 12 ? 12 : MAXK)];     // waves 1-3 park their sums here, MAXK handled in passes of <= 12
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[MAXK], ab[MAXK];
#pragma unroll
  for (int j = 0; j < MAXK; ++j) { ag[j] = 0.f; ab[j] = 0.f; }
  for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
    const float m = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
      const int k = lane + 64 * j;
      if (k < c) {
        const float g = elem<T>::ld(dy + r * c + k) * gamma[k];
        const float xh = (elem<T>::ld(x + r * c + k) - m) * rs;
        s1 += g; s2 += g * xh;
      }
    }
    s1 = wave_sum(s1) / (float)c;
    s2 = wave_sum(s2) / (float)c;
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
      const int k = lane + 64 * j;
      if (k < c) {
        const float d = elem<T>::ld(dy + r * c + k);
        const float xh = (elem<T>::ld(x + r * c + k) - m) * rs;
        elem<T>::st(dx + r * c + k, rs * (d * gamma[k] - s1 - xh * s2));
        ag[j] += d * xh;
        ab[j] += d;
      }
    }
  }
  constexpr int PASS = MAXK > 12 ? 12 : MAXK;
  float *out = partial + (long long)blockIdx.x * 2 * c;
#pragma unroll
  for (int j0 = 0; j0 < MAXK; j0 += PASS) {
    __syncthreads();
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < PASS; ++j)
        if (j0 + j < MAXK) { red[wave - 1][0][j * 64 + lane] = ag[j0 + j]; red[wave - 1][1][j * 64 + lane] = ab[j0 + j]; }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int j = 0; j < PASS; ++j) {
        const int k = lane + 64 * (j0 + j);
        if (j0 + j < MAXK && k < c) {
          out[k] = ag[j0 + j] + red[0][0][j * 64 + lane] + red[1][0][j * 64 + lane] + red[2][0][j * 64 + lane];
          out[c + k] = ab[j0 + j] + red[0][1][j * 64 + lane] + red[1][1][j * 64 + lane] + red[2][1][j * 64 + lane];
        }
      }
    }
  }
}

// dgamma[c], dbeta[c] = sum over the block partials; block = (64 channels, 16 partial lanes)
// acc != 0: the sums are ADDED to dgamma / dbeta (the trainer's gradient-arena slots: no torch add per parameter afterwards)
__global__ void layernorm_param_kernel(const float *__restrict__ partial, int nblocks, int c, float *__restrict__ dgamma,
                                       float *__restrict__ dbeta, int acc) {
  __shared__ float rs[16][64], rq[16][64];
  const int ch = blockIdx.x * 64 + threadIdx.x;
  float a4[4] = {0.f, 0.f, 0.f, 0.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
  if (ch < c) {
    int blk = threadIdx.y;
    for (; blk + 48 < nblocks; blk += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a4[u] += partial[(long long)(blk + 16 * u) * 2 * c + ch];
        b4[u] += partial[(long long)(blk + 16 * u) * 2 * c + c + ch];
      }
    }
    for (; blk < nblocks; blk += 16) { a4[0] += partial[(long long)blk * 2 * c + ch]; b4[0] += partial[(long long)blk * 2 * c + c + ch]; }
  }
  rs[threadIdx.y][threadIdx.x] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  rq[threadIdx.y][threadIdx.x] = (b4[0] + b4[1]) + (b4[2] + b4[3]);
  __syncthreads();
  if (threadIdx.y == 0 && ch < c) {
    float s = 0.f, q = 0.f;
    for (int k = 0; k < 16; ++k) { s += rs[k][threadIdx.x]; q += rq[k][threadIdx.x]; }
    dgamma[ch] = acc ? dgamma[ch] + s : s;
    dbeta[ch] = acc ? dbeta[ch] + q : q;
  }
}

static inline int ln_bwd_blocks(long long rows) { return (int)min((long long)1024, (long long)((rows + 3) / 4)); }

extern "C" size_t nrpn_layernorm_workspace_bytes(int64_t rows, int c) { return (size_t)ln_bwd_blocks(rows) * 2 * c * 4; }

extern "C" int nrpn_layernorm_fwd(const void *x, void *y, const float *gamma, const float *beta, float *mean, float *rstd, int64_t rows,
                                  int c, float eps, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(x && y && gamma && beta && mean && rstd && rows > 0 && c > 0, "layernorm_fwd: bad args");
  const int blocks = (int)min((long long)4096, (long long)((rows + 3) / 4));
  DISPATCH_T(dtype, hipLaunchKernelGGL(layernorm_fwd_kernel<T>, dim3(blocks), dim3(256), 0, as_stream(stream), (const T *)x, (T *)y, gamma, beta,
                                       mean, rstd, (long long)rows, c, eps));
  NRPN_LAUNCH_CHECK("layernorm_fwd");
  return NRPN_OK;
}

extern "C" int nrpn_layernorm_bwd(const void *x, const void *dy, void *dx, const float *gamma, const float *mean, const float *rstd,
                                  float *dgamma, float *dbeta, int64_t rows, int c, int dtype, int accumulate_params, void *workspace,
                                  nrpn_stream_t stream) {
  NRPN_REQUIRE(x && dy && dx && gamma && mean && rstd && dgamma && dbeta && workspace && rows > 0 && c > 0 && c <= 64 * 48,
               "layernorm_bwd: bad args (C <= 3072, workspace = nrpn_layernorm_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  const int blocks = ln_bwd_blocks(rows);
  float *partial = reinterpret_cast<float *>(workspace);
#define NRPN_LNB(K_) DISPATCH_T(dtype, hipLaunchKernelGGL((layernorm_bwd_kernel<T, K_>), dim3(blocks), dim3(256), 0, st, (const T *)x, \
    (const T *)dy, (T *)dx, gamma, mean, rstd, partial, (long long)rows, c))
  if (c <= 64 * 4) { NRPN_LNB(4); } else if (c <= 64 * 12) { NRPN_LNB(12); } else if (c <= 64 * 24) { NRPN_LNB(24); } else { NRPN_LNB(48); }
#undef NRPN_LNB
  hipLaunchKernelGGL(layernorm_param_kernel, dim3((c + 63) / 64), dim3(64, 16), 0, st, (const float *)partial, blocks, c, dgamma, dbeta,
                     accumulate_params ? 1 : 0);
  NRPN_LAUNCH_CHECK("layernorm_bwd");
  return NRPN_OK;
}

// =====================================================================================================================
// exact GELU (nn.GELU default) and the residual join y = a + s[n] * b (s = stochastic-depth row scale, or 1)
// =====================================================================================================================
// 4 elements per thread (8/16-byte accesses) when the count allows it
template <typename T> struct ev4;
template <> struct ev4<float> {
  static __device__ __forceinline__ void ld(const float *p, float *v) { const f4 t = *reinterpret_cast<const f4 *>(p); v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
  static __device__ __forceinline__ void st(float *p, const float *v) { f4 t = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4 *>(p) = t; }
};
template <> struct ev4<bf16s> {
  static __device__ __forceinline__ void ld(const bf16s *p, float *v) {
    const us4 t = *reinterpret_cast<const us4 *>(p);
    v[0] = bf16_bits_to_f32(t[0]); v[1] = bf16_bits_to_f32(t[1]); v[2] = bf16_bits_to_f32(t[2]); v[3] = bf16_bits_to_f32(t[3]);
  }
  static __device__ __forceinline__ void st(bf16s *p, const float *v) {
    us4 t = {f32_to_bf16_bits(v[0]), f32_to_bf16_bits(v[1]), f32_to_bf16_bits(v[2]), f32_to_bf16_bits(v[3])};
    *reinterpret_cast<us4 *>(p) = t;
  }
};

template <typename T, bool BWD>
__global__ void gelu_kernel(const T *__restrict__ x, const T *__restrict__ dy, T *__restrict__ out, long long groups) {
  for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long long)gridDim.x * blockDim.x) {
    float v[4], d[4] = {0.f, 0.f, 0.f, 0.f}, o[4];
    ev4<T>::ld(x + gi * 4, v);
    if (BWD) ev4<T>::ld(dy + gi * 4, d);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float cdf = 0.5f * (1.0f + erff(v[k] * 0.70710678118654752f));
      o[k] = BWD ? d[k] * (cdf + v[k] * 0.39894228040143268f * expf(-0.5f * v[k] * v[k])) : v[k] * cdf;
    }
    ev4<T>::st(out + gi * 4, o);
  }
}

extern "C" int nrpn_gelu(const void *x, const void *dy, void *out, int64_t count, int backward, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(x && out && count > 0 && count % 4 == 0 && (!backward || dy), "gelu: bad args (count must be a multiple of 4)");
  const long long groups = count / 4;
  if (backward) { DISPATCH_T(dtype, hipLaunchKernelGGL((gelu_kernel<T, true>), dim3(ew_blocks(groups)), dim3(256), 0, as_stream(stream),
                                                       (const T *)x, (const T *)dy, (T *)out, groups)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((gelu_kernel<T, false>), dim3(ew_blocks(groups)), dim3(256), 0, as_stream(stream), (const T *)x,
                                              (const T *)nullptr, (T *)out, groups)); }
  NRPN_LAUNCH_CHECK("gelu");
  return NRPN_OK;
}

template <typename T>
__global__ void scale_add_kernel(const T *__restrict__ a, const T *__restrict__ b, const float *__restrict__ scale, T *__restrict__ y,
                                 long long per_sample, long long groups) {
  for (long long gi = (long long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long long)gridDim.x * blockDim.x) {
    const float s = scale ? scale[gi * 4 / per_sample] : 1.0f;
    float av[4] = {0.f, 0.f, 0.f, 0.f}, bv[4], o[4];
    if (a) ev4<T>::ld(a + gi * 4, av);
    ev4<T>::ld(b + gi * 4, bv);
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = av[k] + s * bv[k];
    ev4<T>::st(y + gi * 4, o);
  }
}

extern "C" int nrpn_scale_add(const void *a, const void *b, const float *scale, void *y, int n, int64_t per_sample, int dtype,
                              nrpn_stream_t stream) {
  NRPN_REQUIRE(b && y && n > 0 && per_sample > 0 && per_sample % 4 == 0, "scale_add: bad args (per_sample must be a multiple of 4)");
  const long long groups = (long long)n * per_sample / 4;
  DISPATCH_T(dtype, hipLaunchKernelGGL(scale_add_kernel<T>, dim3(ew_blocks(groups)), dim3(256), 0, as_stream(stream), (const T *)a, (const T *)b,
                                       scale, (T *)y, (long long)per_sample, groups));
  NRPN_LAUNCH_CHECK("scale_add");
  return NRPN_OK;
}

// =====================================================================================================================
// patch merging gather: [N,X,Y,Z,C] -> [N,ceil(X/2),ceil(Y/2),ceil(Z/2), 8C], block q = (x&1) + 2 (y&1) + 4 (z&1);
// odd sizes read zeros (reference F.pad).  The backward is the same index map read the other way.
// =====================================================================================================================
template <typename T, bool FWD>
__global__ void merge_kernel(const T *__restrict__ src, T *__restrict__ dst, int n, int gx, int gy, int gz, int ox, int oy, int oz, int c) {
  const long long total = FWD ? (long long)n * ox * oy * oz * 8 * c : (long long)n * gx * gy * gz * c;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (FWD) {
      const int ch = (int)(i % c);
      long long v = i / c;
      const int q = (int)(v % 8); v /= 8;
      const int z = (int)(v % oz); v /= oz;
      const int y = (int)(v % oy); v /= oy;
      const int x = (int)(v % ox);
      const long long b = v / ox;
      const int sx = 2 * x + (q & 1), sy = 2 * y + ((q >> 1) & 1), sz = 2 * z + (q >> 2);
      const bool in = sx < gx && sy < gy && sz < gz;
      elem<T>::st(dst + i, in ? elem<T>::ld(src + (((b * gx + sx) * gy + sy) * gz + sz) * (long long)c + ch) : 0.f);
    } else {
      const int ch = (int)(i % c);
      long long v = i / c;
      const int z = (int)(v % gz); v /= gz;
      const int y = (int)(v % gy); v /= gy;
      const int x = (int)(v % gx);
      const long long b = v / gx;
      const int q = (x & 1) + 2 * (y & 1) + 4 * (z & 1);
      elem<T>::st(dst + i, elem<T>::ld(src + ((((b * ox + x / 2) * oy + y / 2) * oz + z / 2) * 8 + q) * (long long)c + ch));
    }
  }
}

extern "C" int nrpn_patch_merge(const void *src, void *dst, int n, int gx, int gy, int gz, int c, int backward, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(src && dst && n > 0 && gx > 0 && gy > 0 && gz > 0 && c > 0, "patch_merge: bad args");
  const int ox = (gx + 1) / 2, oy = (gy + 1) / 2, oz = (gz + 1) / 2;
  const long long total = backward ? (long long)n * gx * gy * gz * c : (long long)n * ox * oy * oz * 8 * c;
  if (backward) { DISPATCH_T(dtype, hipLaunchKernelGGL((merge_kernel<T, false>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream),
                                                       (const T *)src, (T *)dst, n, gx, gy, gz, ox, oy, oz, c)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((merge_kernel<T, true>), dim3(ew_blocks(total)), dim3(256), 0, as_stream(stream), (const T *)src,
                                              (T *)dst, n, gx, gy, gz, ox, oy, oz, c)); }
  NRPN_LAUNCH_CHECK("patch_merge");
  return NRPN_OK;
}

// =====================================================================================================================
// shifted-window attention, window 4x4x4 (64 tokens), head_dim 32
// =====================================================================================================================
constexpr int WS = 4, WT = 64, HD = 32;
constexpr int kUnitWs = 408;      // backward workspace floats per (window, head) unit: [0,343) table partial, [344,408) padded-token (k | v) bias partial

struct AttnGeom {
  int n, gx, gy, gz;      // token grid
  int px, py, pz;         // padded to a multiple of the window
  int sx, sy, sz;         // cyclic shift (0 or 2), already zeroed where the window covers the padded axis
  int heads, C;
};

// token `t` of window `w` -> original coordinate; returns false for a padded (zero) token
__device__ __forceinline__ bool attn_token(const AttnGeom &g, int w, int t, long long &row, int &region) {
  const int nwz = g.pz / WS, nwy = g.py / WS, nwx = g.px / WS;
  const int wz = w % nwz, wy = (w / nwz) % nwy, wx = (w / (nwz * nwy)) % nwx, b = w / (nwz * nwy * nwx);
  const int rx = wx * WS + t / 16, ry = wy * WS + (t / 4) % 4, rz = wz * WS + t % 4;      // position in the rolled frame
  const int ox = (rx + g.sx) % g.px, oy = (ry + g.sy) % g.py, oz = (rz + g.sz) % g.pz;   // original (padded) position
  // region ids of the reference's slice loops; an axis that is not shifted ends up with one id (its last slice covers it all)
  const int ax = g.sx == 0 ? 2 : (rx < g.px - WS ? 0 : (rx < g.px - g.sx ? 1 : 2));
  const int ay = g.sy == 0 ? 2 : (ry < g.py - WS ? 0 : (ry < g.py - g.sy ? 1 : 2));
  const int az = g.sz == 0 ? 2 : (rz < g.pz - WS ? 0 : (rz < g.pz - g.sz ? 1 : 2));
  region = (ax * 3 + ay) * 3 + az;
  row = (((long long)b * g.gx + ox) * g.gy + oy) * g.gz + oz;
  return ox < g.gx && oy < g.gy && oz < g.gz;
}

// forward: out[row, head*32 + d] = softmax(q k^T * scale + bias + mask) v
template <typename T>
__global__ void __launch_bounds__(64) window_attn_fwd_kernel(const T *__restrict__ qkv, const float *__restrict__ qkv_bias,
                                                             const float *__restrict__ bias_table, const int *__restrict__ rel_index,
                                                             T *__restrict__ out, AttnGeom g) {
  __shared__ float K[WT][HD + 1], V[WT][HD + 1];
  __shared__ int reg[WT];
  const int w = blockIdx.x, head = blockIdx.y, i = threadIdx.x;
  const bool shifted = (g.sx + g.sy + g.sz) > 0;
  long long row;
  int region;
  const bool real = attn_token(g, w, i, row, region);
  const float scale = 0.17677669529663687f;   // 32^-0.5
  float q[HD];
  const int C = g.C, off = head * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    const float bq = qkv_bias ? qkv_bias[off + d] : 0.f, bk = qkv_bias ? qkv_bias[C + off + d] : 0.f, bv = qkv_bias ? qkv_bias[2 * C + off + d] : 0.f;
    q[d] = (real ? elem<T>::ld(qkv + row * 3 * C + off + d) : bq) * scale;
    K[i][d] = real ? elem<T>::ld(qkv + row * 3 * C + C + off + d) : bk;
    V[i][d] = real ? elem<T>::ld(qkv + row * 3 * C + 2 * C + off + d) : bv;
  }
  reg[i] = region;
  __syncthreads();
  float s[WT];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < WT; ++j) {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) a += q[d] * K[j][d];
    a += bias_table[rel_index[i * WT + j] * g.heads + head];
    if (shifted && reg[j] != region) a += -100.0f;
    s[j] = a;
    mx = fmaxf(mx, a);
  }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < WT; ++j) { s[j] = expf(s[j] - mx); den += s[j]; }
  const float inv = 1.0f / den;
  if (real) {
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      float o = 0.f;
#pragma unroll
      for (int j = 0; j < WT; ++j) o += s[j] * V[j][d];
      elem<T>::st(out + row * C + off + d, o * inv);
    }
  }
}

// backward: recompute P, then dV = P^T dO, dS = P (dP - rowdot), dq = scale dS K, dK = dS^T (scale q), dbias_table += dS
template <typename T>
__global__ void __launch_bounds__(64) window_attn_bwd_kernel(const T *__restrict__ qkv, const float *__restrict__ qkv_bias,
                                                             const float *__restrict__ bias_table, const int *__restrict__ rel_index,
                                                             const T *__restrict__ dout, T *__restrict__ dqkv, float *__restrict__ tabws,
                                                             float *__restrict__ dbias_pad, AttnGeom g) {
  __shared__ float K[WT][HD + 1], V[WT][HD + 1], Q[WT][HD + 1], DO[WT][HD + 1];
  __shared__ float M[WT][WT + 1];
  __shared__ float tab[343], padacc[2 * HD];
  __shared__ int reg[WT];
  const int w = blockIdx.x, head = blockIdx.y, i = threadIdx.x;
  const bool shifted = (g.sx + g.sy + g.sz) > 0;
  long long row;
  int region;
  const bool real = attn_token(g, w, i, row, region);
  const float scale = 0.17677669529663687f;
  const int C = g.C, off = head * HD;
  for (int k = i; k < 343; k += 64) tab[k] = 0.f;
  padacc[i] = 0.f;
  float q[HD], dO[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    const float bq = qkv_bias ? qkv_bias[off + d] : 0.f, bk = qkv_bias ? qkv_bias[C + off + d] : 0.f, bv = qkv_bias ? qkv_bias[2 * C + off + d] : 0.f;
    q[d] = (real ? elem<T>::ld(qkv + row * 3 * C + off + d) : bq) * scale;
    K[i][d] = real ? elem<T>::ld(qkv + row * 3 * C + C + off + d) : bk;
    V[i][d] = real ? elem<T>::ld(qkv + row * 3 * C + 2 * C + off + d) : bv;
    dO[d] = real ? elem<T>::ld(dout + row * C + off + d) : 0.f;     // outputs of padded queries are discarded
    Q[i][d] = q[d];
    DO[i][d] = dO[d];
  }
  reg[i] = region;
  __syncthreads();
  {
    float p[WT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) a += q[d] * K[j][d];
      a += bias_table[rel_index[i * WT + j] * g.heads + head];
      if (shifted && reg[j] != region) a += -100.0f;
      p[j] = a;
      mx = fmaxf(mx, a);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < WT; ++j) { p[j] = expf(p[j] - mx); den += p[j]; }
    const float inv = 1.0f / den;
#pragma unroll
    for (int j = 0; j < WT; ++j) M[i][j] = p[j] * inv;     // P
  }
  __syncthreads();
  // dV_i = sum_r P[r][i] dO_r   (lane i now acts as key/value token i)
  {
    float dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dv[d] = 0.f;
#pragma unroll 2
    for (int r = 0; r < WT; ++r) {
      const float pr = M[r][i];
#pragma unroll
      for (int d = 0; d < HD; ++d) dv[d] += pr * DO[r][d];
    }
    if (real) {
#pragma unroll
      for (int d = 0; d < HD; ++d) elem<T>::st(dqkv + row * 3 * C + 2 * C + off + d, dv[d]);
    } else if (dbias_pad) {   // padded tokens carry the bias vectors as k and v: their gradient goes to the qkv bias (LDS first)
#pragma unroll
      for (int d = 0; d < HD; ++d) atomicAdd(&padacc[HD + d], dv[d]);
    }
  }
  // rowdot_i = sum_j dP_ij P_ij
  float rowdot = 0.f;
#pragma unroll 2
  for (int j = 0; j < WT; ++j) {
    float dp = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) dp += dO[d] * V[j][d];
    rowdot += dp * M[i][j];
  }
  __syncthreads();
  // dS row of query i (overwrites P row i), dq, bias-table gradient
  {
    float dq[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
#pragma unroll 2
    for (int j = 0; j < WT; ++j) {
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) dp += dO[d] * V[j][d];
      const float ds = M[i][j] * (dp - rowdot);
      M[i][j] = ds;
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] += ds * K[j][d];
      atomicAdd(&tab[rel_index[i * WT + j]], ds);
    }
    if (real) {
#pragma unroll
      for (int d = 0; d < HD; ++d) elem<T>::st(dqkv + row * 3 * C + off + d, dq[d] * scale);
    }
  }
  __syncthreads();
  // dK_i = sum_r dS[r][i] * (scale q_r)
  {
    float dk[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) dk[d] = 0.f;
#pragma unroll 2
    for (int r = 0; r < WT; ++r) {
      const float dsr = M[r][i];
#pragma unroll
      for (int d = 0; d < HD; ++d) dk[d] += dsr * Q[r][d];
    }
    if (real) {
#pragma unroll
      for (int d = 0; d < HD; ++d) elem<T>::st(dqkv + row * 3 * C + C + off + d, dk[d]);
    } else if (dbias_pad) {
#pragma unroll
      for (int d = 0; d < HD; ++d) atomicAdd(&padacc[d], dk[d]);
    }
  }
  __syncthreads();
  float *tw = tabws + ((long long)w * g.heads + head) * kUnitWs;   // this unit's partials (summed by attn_table_reduce_kernel / attn_pad_reduce_kernel)
  for (int k = i; k < 343; k += 64) tw[k] = tab[k];
  if (dbias_pad) tw[344 + i] = padacc[i];      // (k | v) bias gradient reaching this unit's padded tokens: plain store, ordered sum later
}

// =====================================================================================================================
// bf16 path: the same attention on the matrix cores.  One wavefront per (window, head); q/k/v/dO rows of the 64 tokens are
// staged once in LDS ([64][32] bf16, 80-byte rows).  Products are formed in the TRANSPOSED orientation
//     S^T[j][i] = sum_d K[j][d] Q[i][d]        (v_mfma_f32_32x32x16_bf16; D layout: lane = query i, registers = keys j)
// so a softmax row lives in one lane pair (l, l ^ 32): 32 in-lane values + one cross-lane exchange.  The probabilities then
// feed the next MFMA straight from registers as its B operand -- the k index of an MFMA is summed out, so A and B only have
// to agree on the key each (lane half h, element e) stands for: j = 16 m + 4 h + (e & 3) + 8 (e >> 2), which is what the D
// layout holds in registers 8m .. 8m+7; the matching A fragment (V^T, K^T, Q^T, dO^T) is two ds_read_b64_tr_b16 of four
// consecutive rows each.  The backward also forms S, dP in the normal orientation (for dV, dK) with the row statistics
// passed through LDS.  Relative-position index = code(i) - code(j) + 171 with code(t) = 49 (t/16) + 7 ((t/4)%4) + t%4
// (the reference's define_relative_position_index for a 4x4x4 window, feature_extractor.py:562-576).
// =====================================================================================================================
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8v;
typedef __attribute__((ext_vector_type(4))) short s4v;
constexpr int AROW = 80;                                    // bytes per staged token row (64 + 16 pad: conflict-free ds_read_b128)

__device__ __forceinline__ int frow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int tok_code(int t) { return (t >> 4) * 49 + ((t >> 2) & 3) * 7 + (t & 3); }
__device__ __forceinline__ f16v mma_bf16(f4 a, f4 b, f16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, a), __builtin_bit_cast(bf8v, b), c, 0, 0, 0);
}
// 8 fp32 -> 8 bf16 packed as an MFMA operand
__device__ __forceinline__ f4 pack8(const float *v) {
  f4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = __uint_as_float((unsigned)f32_to_bf16_bits(v[2 * q]) | ((unsigned)f32_to_bf16_bits(v[2 * q + 1]) << 16));
  return o;
}
// direct fragment: row (tile*32 + lane%32) of a staged tile, 8 consecutive d at kk*16 + (lane/32)*8
__device__ __forceinline__ f4 row_frag(const char *tile, int t, int kk, int lane) {
  return *reinterpret_cast<const f4 *>(tile + (t * 32 + (lane & 31)) * AROW + kk * 32 + (lane >> 5) * 16);
}
// transposed fragment: A[row d = lane%32][k <-> token row0 + {0..3, 8..11}] from a staged [token][d] tile
__device__ __forceinline__ f4 tr_frag(const char *tile, int row0, int lane) {
  const int p = lane & 15, dbase = 16 * ((lane >> 4) & 1);
  const char *a0 = tile + (row0 + (p >> 2)) * AROW + (dbase + 4 * (p & 3)) * 2;
  const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0));
  const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0 + 8 * AROW));
  typedef __attribute__((ext_vector_type(2))) long long l2v;
  l2v r = {__builtin_bit_cast(long long, lo), __builtin_bit_cast(long long, hi)};
  return __builtin_bit_cast(f4, r);
}
// stage the 32 bf16 of token `lane` (or the bias vector for a padded token) into row `lane` of an LDS tile
__device__ __forceinline__ void stage_row(char *tile, int lane, const bf16s *src, const float *bias, bool real) {
  f4 *dst = reinterpret_cast<f4 *>(tile + lane * AROW);
  if (real) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const f4 *>(src + 8 * q);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bias ? bias[8 * q + e] : 0.f;
      dst[q] = pack8(v);
    }
  }
}

struct AttnRows { long long row[2]; bool real[2]; };

// S^T (or dP^T) style product of two staged tiles X (A operand, rows -> D rows) and Y (B operand, rows -> D columns)
__device__ __forceinline__ void tile_product(const char *X, const char *Y, int lane, f16v (&d)[2][2]) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) acc = mma_bf16(row_frag(X, a, kk, lane), row_frag(Y, b, kk, lane), acc);
      d[a][b] = acc;
    }
}

__global__ void __launch_bounds__(64) window_attn_fwd_mfma_kernel(const bf16s *__restrict__ qkv, const float *__restrict__ qkv_bias,
                                                                  const float *__restrict__ bias_table, bf16s *__restrict__ out, AttnGeom g) {
  __shared__ __attribute__((aligned(16))) char Qs[64 * AROW], Ks[64 * AROW], Vs[64 * AROW];
  __shared__ float tabv[343];
  __shared__ int reg[64];
  const int w = blockIdx.x, head = blockIdx.y, lane = threadIdx.x;
  const bool shifted = (g.sx + g.sy + g.sz) > 0;
  const int C = g.C, off = head * HD;
  {
    long long row; int region;
    const bool real = attn_token(g, w, lane, row, region);
    reg[lane] = region;
    const bf16s *src = qkv + row * 3 * C + off;
    stage_row(Qs, lane, src, qkv_bias ? qkv_bias + off : nullptr, real);
    stage_row(Ks, lane, src + C, qkv_bias ? qkv_bias + C + off : nullptr, real);
    stage_row(Vs, lane, src + 2 * C, qkv_bias ? qkv_bias + 2 * C + off : nullptr, real);
    for (int k = lane; k < 343; k += 64) tabv[k] = bias_table[k * g.heads + head];
  }
  __syncthreads();
  const float scale = 0.17677669529663687f;
  f16v st[2][2];                                   // st[tj][ti]: rows = keys j, columns = queries i
  tile_product(Ks, Qs, lane, st);
  float mx[2] = {-INFINITY, -INFINITY}, den[2] = {0.f, 0.f};
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    const int i = ti * 32 + (lane & 31), ci = tok_code(i) + 171, ri = reg[i];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = tj * 32 + frow(r, lane);
        float v = st[tj][ti][r] * scale + tabv[ci - tok_code(j)];
        if (shifted && reg[j] != ri) v += -100.0f;
        st[tj][ti][r] = v;
        mx[ti] = fmaxf(mx[ti], v);
      }
    mx[ti] = fmaxf(mx[ti], __shfl_xor(mx[ti], 32, 64));
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float e = expf(st[tj][ti][r] - mx[ti]); st[tj][ti][r] = e; den[ti] += e; }
    den[ti] += __shfl_xor(den[ti], 32, 64);
  }
  // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) {
    f16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float inv = 1.0f / den[ti];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = st[tj][ti][8 * m + e] * inv;
        acc = mma_bf16(tr_frag(Vs, tj * 32 + 16 * m + 4 * (lane >> 5), lane), pack8(pv), acc);
      }
    long long row; int region;
    if (attn_token(g, w, ti * 32 + (lane & 31), row, region)) {
      bf16s *dst = out + row * C + off;
#pragma unroll
      for (int q = 0; q < 4; ++q) {           // registers 4q..4q+3 are d = 8q + 4h + 0..3: one 8-byte store
        us4 o = {f32_to_bf16_bits(acc[4 * q]), f32_to_bf16_bits(acc[4 * q + 1]), f32_to_bf16_bits(acc[4 * q + 2]), f32_to_bf16_bits(acc[4 * q + 3])};
        *reinterpret_cast<us4 *>(dst + 8 * q + 4 * (lane >> 5)) = o;
      }
    }
  }
}

__global__ void __launch_bounds__(64) window_attn_bwd_mfma_kernel(const bf16s *__restrict__ qkv, const float *__restrict__ qkv_bias,
                                                                  const float *__restrict__ bias_table, const bf16s *__restrict__ dout,
                                                                  bf16s *__restrict__ dqkv, float *__restrict__ tabws,
                                                                  float *__restrict__ dbias_pad, AttnGeom g) {
  __shared__ __attribute__((aligned(16))) char Qs[64 * AROW], Ks[64 * AROW], Vs[64 * AROW], Ds[64 * AROW];
  __shared__ float tabv[343], tabg[343], smax[64], sinv[64], sdot[64], padacc[64];
  __shared__ int reg[64];
  const int w = blockIdx.x, head = blockIdx.y, lane = threadIdx.x, h = lane >> 5;
  const bool shifted = (g.sx + g.sy + g.sz) > 0;
  const int C = g.C, off = head * HD;
  AttnRows tk;                                     // the two tokens (tile 0 / 1) this lane owns as a D-layout column
  {
    long long row; int region;
    const bool real = attn_token(g, w, lane, row, region);
    reg[lane] = region;
    const bf16s *src = qkv + row * 3 * C + off;
    stage_row(Qs, lane, src, qkv_bias ? qkv_bias + off : nullptr, real);
    stage_row(Ks, lane, src + C, qkv_bias ? qkv_bias + C + off : nullptr, real);
    stage_row(Vs, lane, src + 2 * C, qkv_bias ? qkv_bias + 2 * C + off : nullptr, real);
    stage_row(Ds, lane, dout + row * C + off, nullptr, real);            // outputs of padded queries are discarded: dO = 0
    for (int k = lane; k < 343; k += 64) { tabv[k] = bias_table[k * g.heads + head]; tabg[k] = 0.f; }
    padacc[lane] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) { int rg; tk.real[t] = attn_token(g, w, t * 32 + (lane & 31), tk.row[t], rg); }
  }
  __syncthreads();
  const float scale = 0.17677669529663687f;

  // ---------------- transposed orientation: lane = query i, registers = keys j
  {
    f16v st[2][2], dp[2][2];
    tile_product(Ks, Qs, lane, st);                // S^T
    tile_product(Vs, Ds, lane, dp);                // dP^T[j][i] = sum_d V[j][d] dO[i][d]
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
      const int i = ti * 32 + (lane & 31), ci = tok_code(i) + 171, ri = reg[i];
      float mx = -INFINITY, den = 0.f, dot = 0.f;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = tj * 32 + frow(r, lane);
          float v = st[tj][ti][r] * scale + tabv[ci - tok_code(j)];
          if (shifted && reg[j] != ri) v += -100.0f;
          st[tj][ti][r] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float e = expf(st[tj][ti][r] - mx); st[tj][ti][r] = e; den += e; }
      den += __shfl_xor(den, 32, 64);
      const float inv = 1.0f / den;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[tj][ti][r] *= inv; dot += st[tj][ti][r] * dp[tj][ti][r]; }
      dot += __shfl_xor(dot, 32, 64);
      if (h == 0) { smax[i] = mx; sinv[i] = inv; sdot[i] = dot; }
      // dS^T = P^T (dP^T - rowdot); bias-table gradient; dq^T[d][i] = scale sum_j K^T[d][j] dS^T[j][i]
      f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ds = st[tj][ti][r] * (dp[tj][ti][r] - dot);
          dp[tj][ti][r] = ds;
          atomicAdd(&tabg[ci - tok_code(tj * 32 + frow(r, lane))], ds);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float v8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v8[e] = dp[tj][ti][8 * m + e];
          acc = mma_bf16(tr_frag(Ks, tj * 32 + 16 * m + 4 * h, lane), pack8(v8), acc);
        }
      }
      if (tk.real[ti]) {
        bf16s *dst = dqkv + tk.row[ti] * 3 * C + off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          us4 o = {f32_to_bf16_bits(acc[4 * q] * scale), f32_to_bf16_bits(acc[4 * q + 1] * scale), f32_to_bf16_bits(acc[4 * q + 2] * scale),
                   f32_to_bf16_bits(acc[4 * q + 3] * scale)};
          *reinterpret_cast<us4 *>(dst + 8 * q + 4 * h) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---------------- normal orientation: lane = key j, registers = queries i  (P and dS as B operands with k = i)
  {
    f16v sn[2][2], dn[2][2];                       // [ti][tj]
    tile_product(Qs, Ks, lane, sn);                // S[i][j]
    tile_product(Ds, Vs, lane, dn);                // dP[i][j]
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int j = tj * 32 + (lane & 31), cj = tok_code(j), rj = reg[j];
      f16v accv, acck;
#pragma unroll
      for (int r = 0; r < 16; ++r) { accv[r] = 0.f; acck[r] = 0.f; }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = ti * 32 + frow(r, lane);
          float v = sn[ti][tj][r] * scale + tabv[tok_code(i) + 171 - cj];
          if (shifted && reg[i] != rj) v += -100.0f;
          const float p = expf(v - smax[i]) * sinv[i];
          sn[ti][tj][r] = p;
          dn[ti][tj][r] = p * (dn[ti][tj][r] - sdot[i]);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          float p8[8], s8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { p8[e] = sn[ti][tj][8 * m + e]; s8[e] = dn[ti][tj][8 * m + e]; }
          const int row0 = ti * 32 + 16 * m + 4 * h;
          accv = mma_bf16(tr_frag(Ds, row0, lane), pack8(p8), accv);     // dV^T[d][j] += dO^T[d][i] P[i][j]
          acck = mma_bf16(tr_frag(Qs, row0, lane), pack8(s8), acck);     // dK^T[d][j] += Q^T[d][i] dS[i][j]
        }
      }
      if (tk.real[tj]) {
        bf16s *dst = dqkv + tk.row[tj] * 3 * C + off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          us4 ok = {f32_to_bf16_bits(acck[4 * q] * scale), f32_to_bf16_bits(acck[4 * q + 1] * scale), f32_to_bf16_bits(acck[4 * q + 2] * scale),
                    f32_to_bf16_bits(acck[4 * q + 3] * scale)};
          us4 ov = {f32_to_bf16_bits(accv[4 * q]), f32_to_bf16_bits(accv[4 * q + 1]), f32_to_bf16_bits(accv[4 * q + 2]), f32_to_bf16_bits(accv[4 * q + 3])};
          *reinterpret_cast<us4 *>(dst + C + 8 * q + 4 * h) = ok;
          *reinterpret_cast<us4 *>(dst + 2 * C + 8 * q + 4 * h) = ov;
        }
      } else if (dbias_pad) {     // padded tokens carry the bias vectors as k and v: their gradient goes to the qkv bias (LDS first)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = frow(r, lane);
          atomicAdd(&padacc[d], acck[r] * scale);
          atomicAdd(&padacc[32 + d], accv[r]);
        }
      }
    }
  }
  __syncthreads();
  float *tw = tabws + ((long long)w * g.heads + head) * kUnitWs;   // this unit's partials (summed by attn_table_reduce_kernel / attn_pad_reduce_kernel)
  for (int k = lane; k < 343; k += 64) tw[k] = tabg[k];
  if (dbias_pad) tw[344 + lane] = padacc[lane];
}

// dtable[k][head] = sum over the windows' partial tables; block = (64 table entries, 16 window lanes), grid = (6, heads)
__global__ void attn_table_reduce_kernel(const float *__restrict__ tabws, float *__restrict__ dtable, int windows, int heads, int acc) {
  __shared__ float red[16][64];
  const int head = blockIdx.y, k = blockIdx.x * 64 + threadIdx.x;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (k < 343) {
    int w = threadIdx.y;
    for (; w + 48 < windows; w += 64)
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += tabws[((long long)(w + 16 * u) * heads + head) * kUnitWs + k];
    for (; w < windows; w += 16) a[0] += tabws[((long long)w * heads + head) * kUnitWs + k];
  }
  red[threadIdx.y][threadIdx.x] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (threadIdx.y == 0 && k < 343) {
    float sum = 0.f;
    for (int q = 0; q < 16; ++q) sum += red[q][threadIdx.x];
    dtable[k * heads + head] = acc ? dtable[k * heads + head] + sum : sum;      // acc: straight into the gradient-arena slot
  }
}

// dbias_pad[C + head*32 + d] (k half) / [2C + head*32 + d] (v half) = sum over the windows' pad partials, in window order (no atomics);
// block = (64 values, 16 window lanes), grid = heads
__global__ void attn_pad_reduce_kernel(const float *__restrict__ tabws, float *__restrict__ dbias_pad, int windows, int heads, int C) {
  __shared__ float red[16][64];
  const int head = blockIdx.x, i = threadIdx.x;
  float a = 0.f;
  for (int w = threadIdx.y; w < windows; w += 16) a += tabws[((long long)w * heads + head) * kUnitWs + 344 + i];
  red[threadIdx.y][i] = a;
  __syncthreads();
  if (threadIdx.y == 0) {
    float sum = 0.f;
    for (int q = 0; q < 16; ++q) sum += red[q][i];
    dbias_pad[(i < HD ? C : 2 * C) + head * HD + (i & (HD - 1))] = sum;
  }
}

static int g_attn_mfma = 1;   // 1: bf16 tensors use the MFMA kernels (standard relative-position index assumed), 0: always the VALU kernels
extern "C" int nrpn_set_window_attn_mfma(int on) { g_attn_mfma = on ? 1 : 0; return NRPN_OK; }

static int fill_geom(AttnGeom &g, int n, int gx, int gy, int gz, int c, int heads, int shift) {
  if (n <= 0 || gx <= 0 || gy <= 0 || gz <= 0 || heads <= 0 || c != heads * HD)
    return nrpn_fail(NRPN_ERR_ARG, "window_attn: C (%d) must equal heads (%d) * 32", c, heads);
  g.n = n; g.gx = gx; g.gy = gy; g.gz = gz; g.heads = heads; g.C = c;
  g.px = (gx + WS - 1) / WS * WS; g.py = (gy + WS - 1) / WS * WS; g.pz = (gz + WS - 1) / WS * WS;
  g.sx = (shift && WS < g.px) ? WS / 2 : 0;
  g.sy = (shift && WS < g.py) ? WS / 2 : 0;
  g.sz = (shift && WS < g.pz) ? WS / 2 : 0;
  return 0;
}

extern "C" int nrpn_window_attn_fwd(const void *qkv, const float *qkv_bias, const float *bias_table, const int32_t *rel_index, void *out,
                                    int n, int gx, int gy, int gz, int c, int heads, int shift, int dtype, nrpn_stream_t stream) {
  AttnGeom g;
  if (int rc = fill_geom(g, n, gx, gy, gz, c, heads, shift)) return rc;
  NRPN_REQUIRE(qkv && bias_table && rel_index && out, "window_attn_fwd: null pointer");
  dim3 grid((unsigned)(n * (g.px / WS) * (g.py / WS) * (g.pz / WS)), (unsigned)heads);
  if (dtype == NRPN_BF16 && g_attn_mfma) {
    hipLaunchKernelGGL(window_attn_fwd_mfma_kernel, grid, dim3(64), 0, as_stream(stream), (const bf16s *)qkv, qkv_bias, bias_table,
                       (bf16s *)out, g);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(window_attn_fwd_kernel<T>, grid, dim3(64), 0, as_stream(stream), (const T *)qkv, qkv_bias, bias_table,
                                         rel_index, (T *)out, g));
  }
  NRPN_LAUNCH_CHECK("window_attn_fwd");
  return NRPN_OK;
}

extern "C" size_t nrpn_window_attn_bwd_workspace_bytes(int n, int gx, int gy, int gz, int heads) {
  const long long windows = (long long)n * ((gx + WS - 1) / WS) * ((gy + WS - 1) / WS) * ((gz + WS - 1) / WS);
  return (size_t)(windows * heads * kUnitWs * 4);
}

extern "C" int nrpn_window_attn_bwd(const void *qkv, const float *qkv_bias, const float *bias_table, const int32_t *rel_index, const void *dout,
                                    void *dqkv, float *dtable, float *dbias_pad, int n, int gx, int gy, int gz, int c, int heads, int shift,
                                    int dtype, int accumulate_table, void *workspace, nrpn_stream_t stream) {
  AttnGeom g;
  if (int rc = fill_geom(g, n, gx, gy, gz, c, heads, shift)) return rc;
  NRPN_REQUIRE(qkv && bias_table && rel_index && dout && dqkv && dtable && workspace, "window_attn_bwd: null pointer");
  hipStream_t st = as_stream(stream);
  if (dbias_pad) NRPN_HIP(hipMemsetAsync(dbias_pad, 0, (size_t)3 * c * 4, st));
  const int windows = n * (g.px / WS) * (g.py / WS) * (g.pz / WS);
  float *tabws = reinterpret_cast<float *>(workspace);
  dim3 grid((unsigned)windows, (unsigned)heads);
  if (dtype == NRPN_BF16 && g_attn_mfma) {
    hipLaunchKernelGGL(window_attn_bwd_mfma_kernel, grid, dim3(64), 0, st, (const bf16s *)qkv, qkv_bias, bias_table, (const bf16s *)dout,
                       (bf16s *)dqkv, tabws, dbias_pad, g);
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL(window_attn_bwd_kernel<T>, grid, dim3(64), 0, st, (const T *)qkv, qkv_bias, bias_table, rel_index,
                                         (const T *)dout, (T *)dqkv, tabws, dbias_pad, g));
  }
  hipLaunchKernelGGL(attn_table_reduce_kernel, dim3(6, heads), dim3(64, 16), 0, st, (const float *)tabws, dtable, windows, heads,
                     accumulate_table ? 1 : 0);
  if (dbias_pad) hipLaunchKernelGGL(attn_pad_reduce_kernel, dim3(heads), dim3(64, 16), 0, st, (const float *)tabws, dbias_pad, windows, heads, c);
  NRPN_LAUNCH_CHECK("window_attn_bwd");
  return NRPN_OK;
}

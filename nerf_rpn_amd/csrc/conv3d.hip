// Conv3d family for gfx950 (MI355X), channels-last, implicit GEMM on the matrix cores.
//
//   forward / dgrad :  Y[v][n] = sum_{tap,c} X[v + off(tap)][c] * Wp[tap][n][c]          (M = voxels, N = Cout, K = taps*Cin)
//   wgrad           :  dW[tap][n][c] = sum_v dY[v][n] * X[v + off(tap)][c]               (M = Cout,  N = Cin,  K = voxels)
//   stem (Cin = 4, 7^3, stride 1|2): the same two GEMMs with the A (resp. B) operand gathered tap by tap.
//
// One kernel template serves fp32 and bf16: LDS tiles are laid out in BYTES (64-byte K-chunk per row, XOR-swizzled
// 16-byte slots), every fragment is one ds_read_b128, and the 16 bytes feed either one v_mfma_f32_32x32x16_bf16 or four
// v_mfma_f32_32x32x2_f32 (exact fp32, used by the parity path).  Accumulation is always fp32.
// Staging is global -> registers -> LDS, double-buffered, one barrier per K-step; the next K-step's global loads are
// issued before the current step's MFMAs so HBM/L2 latency hides under the matrix work.
//
// Replaces torch.nn.Conv3d (MIOpen/cuDNN) in reference feature_extractor.py:331-358, fpn.py:109-110, anchor.py:190-198.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) short s4v;
typedef unsigned short bf16s;  // raw bf16 storage

template <typename T> struct Mma;
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(f16v &acc, const f4 &a, const f4 &b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
  }
};
template <> struct Mma<bf16s> {
  static __device__ __forceinline__ void run(f16v &acc, const f4 &a, const f4 &b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
  }
};

__device__ __forceinline__ f4 ldg16(const void *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ f4 zero4() { f4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// C/D fragment of the 32x32 MFMA: register r of lane l holds (row, col) = ((r&3) + 8*(r>>2) + 4*(l>>5), l&31)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// =====================================================================================================================
// forward / dgrad
// =====================================================================================================================
struct ConvArgs {
  const void *x;
  const void *w;      // MODE 0: [taps][wrows][Cin];  MODE 1 (stem): [wrows][Kpad], k = tap*4 + c
  const float *bias;
  void *y;
  long long M;        // output voxels (N * OX * OY * OZ)
  int X, Y, Z;        // input grid
  int OX, OY, OZ;     // output grid (== input for MODE 0)
  int Cin, Cout;      // Cout = stored output channels (row length of y)
  int wrows;          // rows per tap in the packed weights (>= Cout; rows >= wrows read as zero)
  int taps;           // 1, 27 (MODE 0) or 343 (MODE 1)
  int stride;         // MODE 1 only
  int flags;
};

template <typename T, int BN, int MODE, bool OUTF32>
__global__ void __launch_bounds__(256) conv_igemm_kernel(const ConvArgs p) {
  constexpr int BM = 128;
  constexpr int WAVES_N = (BN == 128) ? 2 : 1;
  constexpr int TM = (BN == 128) ? 2 : 1;   // 32x32 tiles per wave along M
  constexpr int TN = 2;                      // ... along N
  constexpr int KE = 64 / (int)sizeof(T);    // K elements per step
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  __shared__ __attribute__((aligned(16))) char lds[2 * (A_BYTES + B_BYTES)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int lr = tid >> 2, ls = tid & 3;

  // ---- loader state: two A rows (lr, lr+64) and up to two B rows per thread, one 16-byte slot each
  long long a_off[2];    // element offset of the row's voxel (MODE 0) / unused (MODE 1)
  int a_x[2], a_y[2], a_z[2];
  bool a_ok[2];
  long long a_nbase[2];  // MODE 1: element offset of batch n
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long long v = m0 + lr + 64 * i;
    a_ok[i] = v < p.M;
    const long long vv = a_ok[i] ? v : 0;
    const int oz = (int)(vv % p.OZ);
    const long long t1 = vv / p.OZ;
    const int oy = (int)(t1 % p.OY);
    const long long t2 = t1 / p.OY;
    const int ox = (int)(t2 % p.OX);
    const long long n = t2 / p.OX;
    a_x[i] = ox; a_y[i] = oy; a_z[i] = oz;
    if (MODE == 0) a_off[i] = vv * p.Cin;
    else { a_off[i] = 0; a_nbase[i] = n * (long long)p.X * p.Y * p.Z * 4; }
  }
  constexpr int B_ROWS_PER_THREAD = BN / 64;
  const T *wbase = reinterpret_cast<const T *>(p.w);
  const T *xbase = reinterpret_cast<const T *>(p.x);

  const int cpt = (MODE == 0) ? p.Cin / KE : 1;                                   // K-steps per tap
  const int nk = (MODE == 0) ? p.taps * cpt : (p.taps * 4 + KE - 1) / KE;        // total K-steps
  const int kpad = nk * KE;                                                        // MODE 1 weight row length

  f4 ra[2], rb[B_ROWS_PER_THREAD];

  auto load_step = [&](int ks) {
    if (MODE == 0) {
      const int tap = ks / cpt;
      const int c0 = (ks - tap * cpt) * KE;
      int dx = 0, dy = 0, dz = 0;
      if (p.taps == 27) { dx = tap / 9 - 1; dy = (tap / 3) % 3 - 1; dz = tap % 3 - 1; }
      const long long shift = ((long long)dx * p.Y * p.Z + (long long)dy * p.Z + dz) * p.Cin + c0;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool in = a_ok[i] && (unsigned)(a_x[i] + dx) < (unsigned)p.X && (unsigned)(a_y[i] + dy) < (unsigned)p.Y &&
                        (unsigned)(a_z[i] + dz) < (unsigned)p.Z;
        ra[i] = in ? ldg16(reinterpret_cast<const char *>(xbase + a_off[i] + shift) + ls * 16) : zero4();
      }
#pragma unroll
      for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {
        const int row = n0 + lr + 64 * i;
        rb[i] = (row < p.wrows)
                    ? ldg16(reinterpret_cast<const char *>(wbase + ((long long)tap * p.wrows + row) * p.Cin + c0) + ls * 16)
                    : zero4();
      }
    } else {
      // stem: slot ls of K-step ks covers taps [t0, t0 + TPS), 4 input channels each
      constexpr int TPS = 16 / (4 * (int)sizeof(T));  // taps per 16-byte slot: 1 (fp32) or 2 (bf16)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f4 v = zero4();
#pragma unroll
        for (int q = 0; q < TPS; ++q) {
          const int tap = (ks * 4 + ls) * TPS + q;
          const int dx = tap / 49, dy = (tap / 7) % 7, dz = tap % 7;
          const int ix = a_x[i] * p.stride - 3 + dx, iy = a_y[i] * p.stride - 3 + dy, iz = a_z[i] * p.stride - 3 + dz;
          const bool in = a_ok[i] && tap < p.taps && (unsigned)ix < (unsigned)p.X && (unsigned)iy < (unsigned)p.Y &&
                          (unsigned)iz < (unsigned)p.Z;
          if (in) {
            const T *src = xbase + a_nbase[i] + (((long long)ix * p.Y + iy) * p.Z + iz) * 4;
            if (sizeof(T) == 4) v = ldg16(src);
            else {
              const f2 h = *reinterpret_cast<const f2 *>(src);
              v[2 * q] = h[0]; v[2 * q + 1] = h[1];
            }
          }
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {
        const int row = n0 + lr + 64 * i;
        rb[i] = (row < p.wrows) ? ldg16(reinterpret_cast<const char *>(wbase + (long long)row * kpad + ks * KE) + ls * 16) : zero4();
      }
    }
  };

  auto store_step = [&](int buf) {
    char *A = lds + buf * (A_BYTES + B_BYTES);
    char *B = A + A_BYTES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = lr + 64 * i;
      *reinterpret_cast<f4 *>(A + r * 64 + ((ls ^ ((r >> 2) & 3)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {
      const int r = lr + 64 * i;
      *reinterpret_cast<f4 *>(B + r * 64 + ((ls ^ ((r >> 2) & 3)) << 4)) = rb[i];
    }
  };

  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_step(0);
  store_step(0);
  __syncthreads();

  const int fr = lane & 31, fk = lane >> 5;
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_step(ks + 1);
    const char *A = lds + buf * (A_BYTES + B_BYTES);
    const char *B = A + A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f4 af[TM], bfv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + fr;
        af[i] = *reinterpret_cast<const f4 *>(A + r * 64 + (((s * 2 + fk) ^ ((r >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = (wn * TN + j) * 32 + fr;
        bfv[j] = *reinterpret_cast<const f4 *>(B + r * 64 + (((s * 2 + fk) ^ ((r >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[i], bfv[j]);
    }
    if (ks + 1 < nk) store_step(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: bias / ReLU, store channels-last
  const bool has_bias = (p.flags & NRPN_CONV_BIAS) && p.bias;
  const bool relu = p.flags & NRPN_CONV_RELU;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + fr;
    if (col >= p.Cout) continue;
    const float bv = has_bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long v = m0 + (wm * TM + i) * 32 + frag_row(r, lane);
        if (v < p.M) {
          float o = acc[i][j][r] + bv;
          if (relu) o = fmaxf(o, 0.f);
          if (OUTF32) reinterpret_cast<float *>(p.y)[v * p.Cout + col] = o;
          else elem<T>::st(reinterpret_cast<T *>(p.y) + v * p.Cout + col, o);
        }
      }
    }
  }
}

template <typename T, int MODE>
static int launch_conv(const ConvArgs &a, bool out_f32, hipStream_t st) {
  const int bn = (a.Cout <= 64) ? 64 : 128;
  dim3 grid((unsigned)cdiv64(a.M, 128), (unsigned)((a.Cout + bn - 1) / bn));
#define NRPN_LC(BN_, OF_) hipLaunchKernelGGL((conv_igemm_kernel<T, BN_, MODE, OF_>), grid, dim3(256), 0, st, a)
  if (bn == 64) { if (out_f32) NRPN_LC(64, true); else NRPN_LC(64, false); }
  else { if (out_f32) NRPN_LC(128, true); else NRPN_LC(128, false); }
#undef NRPN_LC
  NRPN_LAUNCH_CHECK("conv_igemm");
  return NRPN_OK;
}

extern "C" int nrpn_conv3d_fwd(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cin,
                               int cout, int wrows, int ksize, int dtype, int flags, nrpn_stream_t stream) {
  NRPN_REQUIRE(ksize == 1 || ksize == 3, "conv3d_fwd: ksize must be 1 or 3 (got %d)", ksize);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "conv3d_fwd: bad dtype %d", dtype);
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0 && cin > 0 && cout > 0 && wrows >= cout, "conv3d_fwd: bad sizes");
  const int es = dtype == NRPN_F32 ? 4 : 2;
  NRPN_REQUIRE((cin * es) % 64 == 0, "conv3d_fwd: Cin*elemsize must be a multiple of 64 bytes (Cin=%d)", cin);
  NRPN_REQUIRE(x && wp && y, "conv3d_fwd: null pointer");
  ConvArgs a{};
  a.x = x; a.w = wp; a.bias = bias; a.y = y;
  a.M = (long long)n * gx * gy * gz;
  a.X = gx; a.Y = gy; a.Z = gz; a.OX = gx; a.OY = gy; a.OZ = gz;   // the kernel splits v into (batch, x, y, z) with these
  a.Cin = cin; a.Cout = cout; a.wrows = wrows; a.taps = ksize == 3 ? 27 : 1; a.stride = 1; a.flags = flags & 3;
  const bool out_f32 = (flags & NRPN_CONV_OUT_F32) != 0;
  if (dtype == NRPN_F32) return launch_conv<float, 0>(a, true, as_stream(stream));
  return launch_conv<bf16s, 0>(a, out_f32, as_stream(stream));
}

extern "C" int nrpn_conv3d_stem_fwd(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cout,
                                    int stride, int dtype, int flags, nrpn_stream_t stream) {
  NRPN_REQUIRE(stride == 1 || stride == 2, "stem: stride must be 1 or 2 (got %d)", stride);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "stem: bad dtype %d", dtype);
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0 && cout > 0, "stem: bad sizes");
  NRPN_REQUIRE(x && wp && y, "stem: null pointer");
  ConvArgs a{};
  a.x = x; a.w = wp; a.bias = bias; a.y = y;
  a.X = gx; a.Y = gy; a.Z = gz;
  a.OX = (gx + 6 - 7) / stride + 1; a.OY = (gy + 6 - 7) / stride + 1; a.OZ = (gz + 6 - 7) / stride + 1;
  a.M = (long long)n * a.OX * a.OY * a.OZ;
  a.Cin = 4; a.Cout = cout; a.wrows = cout; a.taps = 343; a.stride = stride; a.flags = flags & 3;
  if (dtype == NRPN_F32) return launch_conv<float, 1>(a, true, as_stream(stream));
  return launch_conv<bf16s, 1>(a, false, as_stream(stream));
}

// =====================================================================================================================
// wgrad:  dW[tap][m][c] += sum_v dY[v][m] * Xs[v][c]        A = dY^T (M = Cout), B = Xs^T (N = Cin or stem taps*4), K = voxels
// LDS tiles are [voxel][channel] exactly as they sit in memory (row stride RS bytes, padded); the K-major fragments the
// MFMA wants are produced by ds_read_b32 (fp32: one voxel per lane-half) or by the gfx950 transpose read
// ds_read_b64_tr_b16 (bf16: four voxels x one channel per lane).
// =====================================================================================================================
struct WgradArgs {
  const void *x;
  const void *dy;
  float *gw;          // MODE 0: [taps][wrows][Cin]; MODE 1: [wrows][Kpad]
  long long M;        // voxels of dY (N*OX*OY*OZ)
  int X, Y, Z, OX, OY, OZ;
  int Cin, Cout, wrows, taps, stride;
  int ksplit;         // voxel range is cut into ksplit slices (blockIdx.z / taps)
  int kpad;           // MODE 1
  int tr_mode;        // bf16: 1 = ds_read_b64_tr_b16, 0 = scalar 16-bit gathers (slow reference path)
};

template <typename T> struct WgCfg;
template <> struct WgCfg<float> { static constexpr int KV = 32, RS = 128 * 4 + 64; };
template <> struct WgCfg<bf16s> { static constexpr int KV = 32, RS = 128 * 2 + 64; };

// one 32(channel) x 16-byte K fragment out of a [voxel][channel] LDS tile
template <typename T>
__device__ __forceinline__ f4 wg_frag(const char *tile, int ctile0, int kbase, int lane, int tr_mode);

template <>
__device__ __forceinline__ f4 wg_frag<float>(const char *tile, int ctile0, int kbase, int lane, int) {
  // 16 bytes = 4 k-values for lane-half h: voxels kbase + 4h .. +3 (the same permutation on A and B)
  constexpr int RS = WgCfg<float>::RS;
  const int c = ctile0 + (lane & 31), h = lane >> 5;
  f4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float *>(tile + (kbase + 4 * h + q) * RS + c * 4);
  return v;
}

template <>
__device__ __forceinline__ f4 wg_frag<bf16s>(const char *tile, int ctile0, int kbase, int lane, int tr_mode) {
  // 16 bytes = 8 bf16 k-values for lane-half h: voxels kbase + 8h .. +7, channel ctile0 + (lane & 31)
  constexpr int RS = WgCfg<bf16s>::RS;
  const int h = lane >> 5;
  if (tr_mode) {
    // transpose read: inside a 16-lane group, lane i receives element (i & 3) of the 8-byte chunks addressed by lanes
    // 4j + (i >> 2), j = 0..3.  Lane p therefore addresses voxel (p >> 2), channels cbase + 4 (p & 3) .. +3.
    const int p = lane & 15;
    const int cbase = ctile0 + 16 * ((lane >> 4) & 1);
    const int vb = kbase + 8 * h;
    const char *a0 = tile + (vb + (p >> 2)) * RS + (cbase + 4 * (p & 3)) * 2;
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0 + 4 * RS));
    typedef __attribute__((ext_vector_type(8))) short s8v;
    s8v r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(f4, r);
  }
  const int c = ctile0 + (lane & 31);
  typedef __attribute__((ext_vector_type(8))) unsigned short u8v;
  u8v r;
#pragma unroll
  for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const unsigned short *>(tile + (kbase + 8 * h + q) * RS + c * 2);
  return __builtin_bit_cast(f4, r);
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs p) {
  constexpr int KV = WgCfg<T>::KV, RS = WgCfg<T>::RS;
  constexpr int TILE = KV * RS;
  constexpr int PIECES_ROW = 128 * (int)sizeof(T) / 16;       // 16-byte pieces per 128-channel row: 16 (bf16) / 32 (fp32)
  constexpr int PIECES = KV * PIECES_ROW / 256;                // per thread per tile: 2 / 4
  constexpr int KSUB = (sizeof(T) == 2) ? 16 : 8;              // voxels consumed per fragment pair
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 buffers][A tile | B tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 128;   // cout tile
  const int n0 = blockIdx.y * 128;   // cin tile (MODE 0) / k tile (MODE 1)
  const int tap = (MODE == 0) ? (int)(blockIdx.z % p.taps) : 0;
  const int slice = (MODE == 0) ? (int)(blockIdx.z / p.taps) : (int)blockIdx.z;
  int dx = 0, dy = 0, dz = 0;
  if (MODE == 0 && p.taps == 27) { dx = tap / 9 - 1; dy = (tap / 3) % 3 - 1; dz = tap % 3 - 1; }

  const long long chunks = (p.M + KV - 1) / KV;
  const long long per = (chunks + p.ksplit - 1) / p.ksplit;
  const long long c_begin = slice * per, c_end = min(chunks, c_begin + per);
  if (c_begin >= c_end) return;

  const T *xbase = reinterpret_cast<const T *>(p.x);
  const T *dybase = reinterpret_cast<const T *>(p.dy);

  f4 ra[PIECES], rb[PIECES];

  auto load_chunk = [&](long long ch) {
    const long long v0 = ch * KV;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i;
      const int row = pc / PIECES_ROW, col = pc % PIECES_ROW;   // voxel row in the chunk, 16-byte column
      const long long v = v0 + row;
      const bool vok = v < p.M;
      // A: dY[v][m0 + ...]
      {
        const int c0 = m0 + col * (16 / (int)sizeof(T));
        ra[i] = (vok && c0 < p.Cout) ? ldg16(dybase + v * p.Cout + c0) : zero4();
      }
      // B
      if (MODE == 0) {
        const int c0 = n0 + col * (16 / (int)sizeof(T));
        bool in = vok && c0 < p.Cin;
        long long src = 0;
        if (in) {
          const int z = (int)(v % p.Z);
          const long long t1 = v / p.Z;
          const int y = (int)(t1 % p.Y);
          const int x = (int)((t1 / p.Y) % p.X);
          in = (unsigned)(x + dx) < (unsigned)p.X && (unsigned)(y + dy) < (unsigned)p.Y && (unsigned)(z + dz) < (unsigned)p.Z;
          src = (v + ((long long)dx * p.Y + dy) * p.Z + dz) * p.Cin + c0;
        }
        rb[i] = in ? ldg16(xbase + src) : zero4();
      } else {
        // stem: column = taps [t0, t0 + TPS) x 4 channels of the im2col row of output voxel v
        constexpr int TPS = 16 / (4 * (int)sizeof(T));
        f4 val = zero4();
        if (vok) {
          const int oz = (int)(v % p.OZ);
          const long long t1 = v / p.OZ;
          const int oy = (int)(t1 % p.OY);
          const long long t2 = t1 / p.OY;
          const int ox = (int)(t2 % p.OX);
          const long long nb = (t2 / p.OX) * (long long)p.X * p.Y * p.Z * 4;
#pragma unroll
          for (int q = 0; q < TPS; ++q) {
            const int tp = (n0 / 4) + col * TPS + q;
            const int tx = tp / 49, ty = (tp / 7) % 7, tz = tp % 7;
            const int ix = ox * p.stride - 3 + tx, iy = oy * p.stride - 3 + ty, iz = oz * p.stride - 3 + tz;
            if (tp < p.taps && (unsigned)ix < (unsigned)p.X && (unsigned)iy < (unsigned)p.Y && (unsigned)iz < (unsigned)p.Z) {
              const T *src = xbase + nb + (((long long)ix * p.Y + iy) * p.Z + iz) * 4;
              if (sizeof(T) == 4) val = ldg16(src);
              else { const f2 h2 = *reinterpret_cast<const f2 *>(src); val[2 * q] = h2[0]; val[2 * q + 1] = h2[1]; }
            }
          }
        }
        rb[i] = val;
      }
    }
  };

  auto store_chunk = [&](int buf) {
    char *A = lds + buf * 2 * TILE;
    char *B = A + TILE;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i;
      const int row = pc / PIECES_ROW, col = pc % PIECES_ROW;
      *reinterpret_cast<f4 *>(A + row * RS + col * 16) = ra[i];
      *reinterpret_cast<f4 *>(B + row * RS + col * 16) = rb[i];
    }
  };

  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_chunk(c_begin);
  store_chunk(0);
  __syncthreads();
  int buf = 0;
  for (long long ch = c_begin; ch < c_end; ++ch) {
    if (ch + 1 < c_end) load_chunk(ch + 1);
    const char *A = lds + buf * 2 * TILE;
    const char *B = A + TILE;
#pragma unroll
    for (int kb = 0; kb < KV; kb += KSUB) {
      f4 af[2], bfv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = wg_frag<T>(A, (wm * 2 + i) * 32, kb, lane, p.tr_mode);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfv[j] = wg_frag<T>(B, (wn * 2 + j) * 32, kb, lane, p.tr_mode);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], af[i], bfv[j]);
    }
    if (ch + 1 < c_end) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  const int fr = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + (wn * 2 + j) * 32 + fr;
    const int ncols = (MODE == 0) ? p.Cin : p.kpad;
    if (col >= ncols) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * 2 + i) * 32 + frag_row(r, lane);
        if (row < p.wrows) {
          float *dst = (MODE == 0) ? p.gw + ((long long)tap * p.wrows + row) * p.Cin + col : p.gw + (long long)row * p.kpad + col;
          atomicAdd(dst, acc[i][j][r]);
        }
      }
  }
}

static int g_wgrad_tr_mode = 1;
extern "C" int nrpn_set_wgrad_transpose_read(int on) { g_wgrad_tr_mode = on ? 1 : 0; return NRPN_OK; }

template <typename T, int MODE>
static int launch_wgrad(WgradArgs a, int ntiles_n, hipStream_t st) {
  const int tiles = ((a.wrows + 127) / 128) * ntiles_n * (MODE == 0 ? a.taps : 1);
  const long long chunks = (a.M + 31) / 32;
  long long ks = (1024 + tiles - 1) / tiles;
  if (ks > chunks / 4) ks = chunks / 4;
  if (ks < 1) ks = 1;
  if (ks > 4096) ks = 4096;
  a.ksplit = (int)ks;
  a.tr_mode = g_wgrad_tr_mode;
  const size_t lds = 4 * (size_t)WgCfg<T>::KV * WgCfg<T>::RS;
  static bool attr_done[2][2] = {{false, false}, {false, false}};
  bool &done = attr_done[sizeof(T) == 2][MODE];
  if (!done) {
    NRPN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_kernel<T, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    done = true;
  }
  dim3 grid((a.wrows + 127) / 128, ntiles_n, (MODE == 0 ? a.taps : 1) * a.ksplit);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, MODE>), grid, dim3(256), lds, st, a);
  NRPN_LAUNCH_CHECK("conv_wgrad");
  return NRPN_OK;
}

// per-channel column sums of dY (bias gradient): rows x C -> C
template <typename T>
__global__ void colsum_kernel(const T *__restrict__ dy, long long rows, int c, float *__restrict__ out) {
  // blockDim = (64, 4): x strides channels, y strides rows inside the block's row slab
  const long long per = (rows + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * per, r1 = min(rows, r0 + per);
  for (int ch = threadIdx.x; ch < c; ch += 64) {
    float s = 0.f;
    for (long long r = r0 + threadIdx.y; r < r1; r += 4) s += elem<T>::ld(dy + r * c + ch);
    atomicAdd(out + ch, s);
  }
}

extern "C" int nrpn_colsum(const void *dy, long long rows, int c, int dtype, float *out, nrpn_stream_t stream) {
  NRPN_REQUIRE(dy && out && rows > 0 && c > 0, "colsum: bad args");
  hipStream_t st = as_stream(stream);
  NRPN_HIP(hipMemsetAsync(out, 0, (size_t)c * 4, st));
  const int blocks = (int)min((long long)1024, (rows + 63) / 64);
  if (dtype == NRPN_F32) hipLaunchKernelGGL(colsum_kernel<float>, dim3(blocks), dim3(64, 4), 0, st, (const float *)dy, rows, c, out);
  else hipLaunchKernelGGL(colsum_kernel<bf16s>, dim3(blocks), dim3(64, 4), 0, st, (const bf16s *)dy, rows, c, out);
  NRPN_LAUNCH_CHECK("colsum");
  return NRPN_OK;
}

extern "C" int nrpn_conv3d_wgrad(const void *x, const void *dy, float *gw_packed, float *gbias, int n, int gx, int gy, int gz, int cin,
                                 int cout, int wrows, int ksize, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(ksize == 1 || ksize == 3, "conv3d_wgrad: ksize must be 1 or 3 (got %d)", ksize);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "conv3d_wgrad: bad dtype %d", dtype);
  const int es = dtype == NRPN_F32 ? 4 : 2;
  NRPN_REQUIRE((cin * es) % 16 == 0 && (cout * es) % 16 == 0, "conv3d_wgrad: channel rows must be 16-byte multiples (Cin=%d Cout=%d)", cin, cout);
  NRPN_REQUIRE(x && dy && gw_packed && wrows >= cout, "conv3d_wgrad: bad args");
  WgradArgs a{};
  a.x = x; a.dy = dy; a.gw = gw_packed;
  a.M = (long long)n * gx * gy * gz;
  a.X = gx; a.Y = gy; a.Z = gz; a.OX = gx; a.OY = gy; a.OZ = gz;
  a.Cin = cin; a.Cout = cout; a.wrows = wrows; a.taps = ksize == 3 ? 27 : 1; a.stride = 1; a.kpad = 0;
  hipStream_t st = as_stream(stream);
  NRPN_HIP(hipMemsetAsync(gw_packed, 0, (size_t)a.taps * wrows * cin * 4, st));
  int rc;
  if (dtype == NRPN_F32) rc = launch_wgrad<float, 0>(a, (cin + 127) / 128, st);
  else rc = launch_wgrad<bf16s, 0>(a, (cin + 127) / 128, st);
  if (rc) return rc;
  if (gbias) return nrpn_colsum(dy, a.M, cout, dtype, gbias, stream);
  return NRPN_OK;
}

extern "C" int nrpn_conv3d_stem_wgrad(const void *x, const void *dy, float *gw_packed, float *gbias, int n, int gx, int gy, int gz,
                                      int cout, int stride, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(stride == 1 || stride == 2, "stem wgrad: stride must be 1 or 2 (got %d)", stride);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "stem wgrad: bad dtype %d", dtype);
  const int es = dtype == NRPN_F32 ? 4 : 2;
  NRPN_REQUIRE((cout * es) % 16 == 0, "stem wgrad: Cout row must be a 16-byte multiple");
  NRPN_REQUIRE(x && dy && gw_packed, "stem wgrad: null pointer");
  WgradArgs a{};
  a.x = x; a.dy = dy; a.gw = gw_packed;
  a.X = gx; a.Y = gy; a.Z = gz;
  a.OX = (gx - 1) / stride + 1; a.OY = (gy - 1) / stride + 1; a.OZ = (gz - 1) / stride + 1;
  a.M = (long long)n * a.OX * a.OY * a.OZ;
  a.Cin = 4; a.Cout = cout; a.wrows = cout; a.taps = 343; a.stride = stride;
  const int ke = 64 / es;
  a.kpad = ((343 * 4 + ke - 1) / ke) * ke;
  hipStream_t st = as_stream(stream);
  NRPN_HIP(hipMemsetAsync(gw_packed, 0, (size_t)cout * a.kpad * 4, st));
  int rc;
  if (dtype == NRPN_F32) rc = launch_wgrad<float, 1>(a, (a.kpad + 127) / 128, st);
  else rc = launch_wgrad<bf16s, 1>(a, (a.kpad + 127) / 128, st);
  if (rc) return rc;
  if (gbias) return nrpn_colsum(dy, a.M, cout, dtype, gbias, stream);
  return NRPN_OK;
}

// =====================================================================================================================
// weight packing (reference layout [Cout][Cin][taps] fp32  <->  GEMM layouts)
// =====================================================================================================================
template <typename T>
__global__ void pack_weight_kernel(const float *__restrict__ w, int cout, int cin, int taps, T *__restrict__ fwd, T *__restrict__ dgrad,
                                   int rows_total, int row_offset) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)cout * cin * taps;
  if (i >= total) return;
  const int t = (int)(i % taps);
  const int c = (int)((i / taps) % cin);
  const int o = (int)(i / ((long long)taps * cin));
  const float v = w[i];
  if (fwd) elem<T>::st(fwd + ((long long)t * rows_total + row_offset + o) * cin + c, v);
  if (dgrad) elem<T>::st(dgrad + ((long long)(taps - 1 - t) * cin + c) * rows_total + row_offset + o, v);
}

extern "C" int nrpn_pack_conv_weight(const float *w_ref, int cout, int cin, int taps, int dtype, void *wp_fwd, void *wp_dgrad,
                                     int rows_total, int row_offset, nrpn_stream_t stream) {
  NRPN_REQUIRE(w_ref && (wp_fwd || wp_dgrad) && cout > 0 && cin > 0 && taps > 0 && row_offset >= 0 && row_offset + cout <= rows_total,
               "pack_conv_weight: bad args");
  const long long total = (long long)cout * cin * taps;
  dim3 grid((unsigned)cdiv64(total, 256));
  if (dtype == NRPN_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, grid, dim3(256), 0, as_stream(stream), w_ref, cout, cin, taps, (float *)wp_fwd,
                       (float *)wp_dgrad, rows_total, row_offset);
  else
    hipLaunchKernelGGL(pack_weight_kernel<bf16s>, grid, dim3(256), 0, as_stream(stream), w_ref, cout, cin, taps, (bf16s *)wp_fwd,
                       (bf16s *)wp_dgrad, rows_total, row_offset);
  NRPN_LAUNCH_CHECK("pack_conv_weight");
  return NRPN_OK;
}

__global__ void unpack_wgrad_kernel(const float *__restrict__ gp, int cout, int cin, int taps, int rows_total, int row_offset,
                                    float *__restrict__ gw, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)cout * cin * taps;
  if (i >= total) return;
  const int t = (int)(i % taps);
  const int c = (int)((i / taps) % cin);
  const int o = (int)(i / ((long long)taps * cin));
  const float v = gp[((long long)t * rows_total + row_offset + o) * cin + c];
  gw[i] = accumulate ? gw[i] + v : v;
}

extern "C" int nrpn_unpack_conv_wgrad(const float *gw_packed, int cout, int cin, int taps, int rows_total, int row_offset, float *gw_ref,
                                      int accumulate, nrpn_stream_t stream) {
  NRPN_REQUIRE(gw_packed && gw_ref && cout > 0 && cin > 0 && taps > 0 && row_offset + cout <= rows_total, "unpack_conv_wgrad: bad args");
  const long long total = (long long)cout * cin * taps;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, as_stream(stream), gw_packed, cout, cin, taps,
                     rows_total, row_offset, gw_ref, accumulate);
  NRPN_LAUNCH_CHECK("unpack_conv_wgrad");
  return NRPN_OK;
}

// stem weights: reference [Cout][4][343] fp32 -> [Cout][Kpad] with k = tap*4 + c (zero padded), and back for the gradient
template <typename T>
__global__ void pack_stem_kernel(const float *__restrict__ w, int cout, int kpad, T *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)cout * kpad) return;
  const int k = (int)(i % kpad), o = (int)(i / kpad);
  const int tap = k >> 2, c = k & 3;
  elem<T>::st(out + i, tap < 343 ? w[((long long)o * 4 + c) * 343 + tap] : 0.f);
}

extern "C" int nrpn_stem_kpad(int dtype) { const int ke = dtype == NRPN_F32 ? 16 : 32; return ((343 * 4 + ke - 1) / ke) * ke; }

extern "C" int nrpn_pack_stem_weight(const float *w_ref, int cout, int dtype, void *wp, nrpn_stream_t stream) {
  NRPN_REQUIRE(w_ref && wp && cout > 0, "pack_stem_weight: bad args");
  const int kpad = nrpn_stem_kpad(dtype);
  dim3 grid((unsigned)cdiv64((long long)cout * kpad, 256));
  if (dtype == NRPN_F32) hipLaunchKernelGGL(pack_stem_kernel<float>, grid, dim3(256), 0, as_stream(stream), w_ref, cout, kpad, (float *)wp);
  else hipLaunchKernelGGL(pack_stem_kernel<bf16s>, grid, dim3(256), 0, as_stream(stream), w_ref, cout, kpad, (bf16s *)wp);
  NRPN_LAUNCH_CHECK("pack_stem_weight");
  return NRPN_OK;
}

__global__ void unpack_stem_wgrad_kernel(const float *__restrict__ gp, int cout, int kpad, float *__restrict__ gw, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)cout * 4 * 343) return;
  const int tap = (int)(i % 343), c = (int)((i / 343) % 4), o = (int)(i / (343 * 4));
  const float v = gp[(long long)o * kpad + tap * 4 + c];
  gw[i] = accumulate ? gw[i] + v : v;
}

extern "C" int nrpn_unpack_stem_wgrad(const float *gw_packed, int cout, int dtype, float *gw_ref, int accumulate, nrpn_stream_t stream) {
  NRPN_REQUIRE(gw_packed && gw_ref && cout > 0, "unpack_stem_wgrad: bad args");
  const int kpad = nrpn_stem_kpad(dtype);
  hipLaunchKernelGGL(unpack_stem_wgrad_kernel, dim3((unsigned)cdiv64((long long)cout * 4 * 343, 256)), dim3(256), 0, as_stream(stream),
                     gw_packed, cout, kpad, gw_ref, accumulate);
  NRPN_LAUNCH_CHECK("unpack_stem_wgrad");
  return NRPN_OK;
}

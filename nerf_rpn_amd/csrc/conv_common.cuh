// Shared device helpers and argument structs of the conv3d translation units (conv3d.hip, conv_halo.hip).
#pragma once
#include "common.h"

#include <atomic>

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
// 16-byte raw-buffer load: a lane whose offset is >= the descriptor's num_records gets zeros from the hardware, so border
// taps / tail rows need no branch and no select -- every load of a K-step issues back to back.
typedef __attribute__((ext_vector_type(4))) unsigned int u4v;
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f4 bufld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ f4 zero4() { f4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// XCD-aware workgroup order.  The dispatcher places linear workgroup id b on XCD b % 8 (observed, speed only); each XCD has
// a private 4 MiB L2.  Remapping id -> (id % 8) * ceil(n/8) + id / 8 hands every XCD one CONTIGUOUS range of logical tiles,
// so neighbouring tiles (which share halo voxels / weight panels / K-slices) hit the same L2 instead of eight different ones.
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n) {
  const unsigned q = n / 8, r = n % 8, xcd = id % 8, slot = id / 8;
  // bijective for any n: the first r XCDs own q+1 ids, the rest q
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// 16-byte LDS-DMA: each lane's 16 bytes land at (wave-uniform lds_dst) + 16 * lane; out-of-range lanes deposit zeros
// (verified on hardware by tools/probe_glds.hip)
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t r, void *lds_dst, unsigned voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff, 0, 0, 0);
}

// C/D fragment of the 32x32 MFMA: register r of lane l holds (row, col) = ((r&3) + 8*(r>>2) + 4*(l>>5), l&31)
__device__ __forceinline__ int frag_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// =====================================================================================================================
// forward / dgrad
// =====================================================================================================================
// Ragged voxel lists: several grids laid end to end ((level, scene) segments of a weight-sharing head run in ONE launch).
// n == 0 means the classic layout: N copies of one X*Y*Z grid.
constexpr int kMaxSeg = 16;
struct Segs {
  int n;
  int start[kMaxSeg + 1];                 // first voxel of each segment, start[n] = total
  int X[kMaxSeg], Y[kMaxSeg], Z[kMaxSeg];
};

// voxel -> coordinates inside its grid, that grid's dims and its segment id (no dynamic indexing of the kernel-argument struct)
__device__ __forceinline__ int locate_voxel(const Segs &s, long long v, int cX, int cY, int cZ, int &x, int &y, int &z, int &X, int &Y,
                                            int &Z) {
  int seg = 0;
  long long local = v;
  if (s.n > 0) {
    X = s.X[0]; Y = s.Y[0]; Z = s.Z[0];
    int st = 0;
#pragma unroll
    for (int k = 1; k < kMaxSeg; ++k)
      if (k < s.n && v >= s.start[k]) { X = s.X[k]; Y = s.Y[k]; Z = s.Z[k]; st = s.start[k]; seg = k; }
    local = v - st;
  } else {
    X = cX; Y = cY; Z = cZ;
  }
  z = (int)(local % Z);
  const long long t1 = local / Z;
  y = (int)(t1 % Y);
  x = (int)((t1 / Y) % X);
  return seg;
}

// FastDiv / make_fastdiv / fastdiv: common.h (division by a launch-invariant divisor as a multiply-shift; the loaders' prologue uses it for the
// voxel decomposition -- ~1 600 dynamic VALU instructions per wave before the first load with three 64-bit divisions per tile row).
// 27-bit in-bounds mask of the 3x3x3 taps of voxel (x, y, z) in an X x Y x Z grid, bit t = (dx+1) * 9 + (dy+1) * 3 + (dz+1): three 3-bit axis
// masks combined with shifts instead of 27 x 3 compares
__device__ __forceinline__ unsigned tap_mask27(int x, int y, int z, int X, int Y, int Z) {
  const unsigned mz = (z > 0 ? 1u : 0u) | 2u | (z + 1 < Z ? 4u : 0u);
  const unsigned my = (y > 0 ? 1u : 0u) | 2u | (y + 1 < Y ? 4u : 0u);
  const unsigned m9 = (my & 1u ? mz : 0u) | (my & 2u ? mz << 3 : 0u) | (my & 4u ? mz << 6 : 0u);
  return (x > 0 ? m9 : 0u) | (m9 << 9) | (x + 1 < X ? m9 << 18 : 0u);
}

static inline int fill_segs(Segs &sg, int nseg, const int32_t *dims, long long &M) {
  if (nseg < 1 || nseg > kMaxSeg || !dims) return nrpn_fail(NRPN_ERR_ARG, "ragged conv: 1..%d segments", kMaxSeg);
  sg.n = nseg;
  long long off = 0;
  for (int k = 0; k < nseg; ++k) {
    if (dims[3 * k] <= 0 || dims[3 * k + 1] <= 0 || dims[3 * k + 2] <= 0) return nrpn_fail(NRPN_ERR_ARG, "ragged conv: bad segment %d", k);
    sg.start[k] = (int)off; sg.X[k] = dims[3 * k]; sg.Y[k] = dims[3 * k + 1]; sg.Z[k] = dims[3 * k + 2];
    off += (long long)dims[3 * k] * dims[3 * k + 1] * dims[3 * k + 2];
    if (off >= (1ll << 31)) return nrpn_fail(NRPN_ERR_ARG, "ragged conv: too many voxels");
  }
  for (int k = nseg; k <= kMaxSeg; ++k) sg.start[k] = (int)off;
  M = off;
  return 0;
}

struct ConvArgs {
  const void *x;
  const void *w;      // MODE 0: [taps][wrows][Cin];  MODE 1 (stem): [wrows][Kpad], k = tap*4 + c
  const float *bias;
  const float *scale; // optional f32 [Cout]: y = acc * scale + bias (eval-mode BatchNorm folded into the conv: scale = gamma / sqrt(var + eps),
                      // bias = (conv bias - mean) * scale + beta); nullptr = 1
  const void *mask;   // optional [M][Cout] (dtype of x): outputs are zeroed where mask <= 0 (ReLU backward of the tensor this dgrad feeds)
  void *y;
  long long M;        // output voxels (N * OX * OY * OZ)
  int X, Y, Z;        // input grid
  int OX, OY, OZ;     // output grid (== input for MODE 0)
  int Cin, Cout;      // Cout = stored output channels (row length of y)
  int wrows;          // rows per tap in the packed weights (>= Cout; rows >= wrows read as zero)
  int taps;           // 1, 27 (MODE 0) or 343 (MODE 1)
  int stride;         // MODE 1 only
  int flags;
  unsigned x_bytes, w_bytes;   // extents for the raw-buffer descriptors (out-of-range lanes read 0)
  int ksplit;         // > 1: workgroup id / tiles owns a slice of the K loop and stores its fp32 partial into ws[slice][M][Cout]
  float *ws;          // [ksplit][M][Cout] fp32 partials (plain stores, summed in slice order by splitk_epilogue_kernel: deterministic)
  int slices;         // 1: the K slices run on the 256x256 kernel (mid-size grids, conv_big_split); 0: on the 128-row kernel
  Segs segs;          // MODE 0 only: ragged voxel list (n > 0) instead of N copies of X*Y*Z
  int tail_tile0, tail_ks;   // conv_igemm_big_kernel: tail_ks > 1 = the M tiles from tail_tile0 on run on tail_ks K slices (partials for rows >=
                             // tail_tile0 * 256 in ws), the tiles before them whole; 0 = off
  const unsigned *rows;   // row-list form (conv_igemm_kernel<..., ROWS = true> only): [M][2] u32 = {voxel id, tap word}, see csrc/cone.hip
  FastDiv dvz, dvy, dvs;   // classic layout (segs.n == 0): divisions by Z, Y and X*Y*Z (voxels per scene) of the loaders' prologue
  float *stats;       // optional (bf16 staged epilogues only): per row-group partial BatchNorm statistics [P][2][Cout] = (sum, sum of squares)
                      // of the STORED (bf16-rounded) outputs; row group = the rows one wave row covers (see nrpn_conv3d_fwd_stats_rows)
};

// per-column (sum, sum of squares) of the values a lane holds in its C fragments -> partial statistics row `pidx` (lanes l and l^32 hold
// the two row halves of the same 32 columns)
__device__ __forceinline__ void store_col_stats(float *stats, long long pidx, int cout, int col, float s, float q, int lane) {
  s += __shfl_xor(s, 32, 64);
  q += __shfl_xor(q, 32, 64);
  if (lane < 32 && col < cout) {
    stats[(pidx * 2) * cout + col] = s;
    stats[(pidx * 2 + 1) * cout + col] = q;
  }
}


// RPN post-processing and target-assignment kernels (gfx950): rotated IoU, bitmask 3D NMS, segmented top-k,
// anchor/coder kernels, proposal filter, matcher, sampled losses.  Rows a8-a20 of SURVEY.md section 8a.
// These stages are scan/sort/IoU-type (HBM- or latency-bound), not GEMMs: one lane per box (pair), wave-wide
// ballots/shuffles for reductions, LDS for the per-workgroup sort and the NMS row blocks.
#include "geometry.cuh"

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>

thread_local char g_nrpn_err[512] = "";

int nrpn_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_nrpn_err, sizeof(g_nrpn_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *nrpn_last_error(void) { return g_nrpn_err; }

int nrpn_ensure_dynamic_lds(const void *kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, int> granted;      // (kernel, device) -> bytes already granted
  int dev = 0;
  NRPN_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  int &have = granted[{kernel, dev}];
  if (have >= bytes) return NRPN_OK;
  NRPN_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
  return NRPN_OK;
}
extern "C" int nrpn_abi_version(void) { return 3; }
// tools / tests: the multiply-shift division of common.h (FastDiv) evaluated on the HOST with the same magic pair and the same arithmetic the
// kernels use (mulhi, shift) -- lets the CPU suite check the construction against n / d without a GPU
extern "C" int64_t nrpn_fastdiv_host(int64_t n, int64_t d) {
  if (n < 0 || n >= (1ll << 31) || d < 1 || d >= (1ll << 31)) return -1;
  const FastDiv f = make_fastdiv((unsigned)d);
  if (f.d == 1) return n;
  return (int64_t)((unsigned)(((unsigned long long)(unsigned)n * f.m) >> 32) >> f.sh);
}

extern "C" int nrpn_check_device(int ordinal) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, ordinal) != hipSuccess) return nrpn_fail(NRPN_ERR_DEVICE, "no HIP device %d", ordinal);
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
    return nrpn_fail(NRPN_ERR_DEVICE, "device %d is %s, this library is built for gfx950 only", ordinal, p.gcnArchName);
  return NRPN_OK;
}

// =====================================================================================================================
// sort_vertices drop-in (one lane per polygon; the reference launches <<<B, pow2<=512>>> with a per-thread stride loop,
// i.e. a single block when B == 1 -- here the grid is flat over B*N).
// =====================================================================================================================
__global__ void sort_vertices_kernel(const float *__restrict__ v, const uint8_t *__restrict__ msk,
                                     const int32_t *__restrict__ num_valid, int32_t *__restrict__ idx, int64_t bn, int m) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bn) return;
  v += i * m * 2;
  msk += i * m;
  int32_t *out = idx + i * 9;
  int nv = num_valid[i];
  int pad = m - 1;
  for (int j = 8; j < m; ++j)
    if (!msk[j]) { pad = j; break; }
  if (nv < 3) {
    for (int j = 0; j < 9; ++j) out[j] = pad;
    return;
  }
  if (nv > 8) nv = 8;
  int loc[9];
  float px = 0.f, py = 0.f;
  for (int j = 0; j < nv; ++j) {
    float bx = 1.0f, by = (float)(-1e-8);
    int take = 0;
    for (int k = 0; k < m; ++k) {
      if (!msk[k]) continue;
      const float x = v[2 * k], y = v[2 * k + 1];
      bool c = geo::vert_before(x, y, bx, by);
      if (j > 0) c = c && geo::vert_before(px, py, x, y);
      if (c) { bx = x; by = y; take = k; }
    }
    loc[j] = take;
    px = v[2 * take];
    py = v[2 * take + 1];
  }
  loc[nv] = loc[0];
  for (int j = nv + 1; j < 9; ++j) loc[j] = pad;
  if (nv == 8) {
    int dup = 0;
    for (int j = 0; j < 4; ++j)
      for (int k = 4; k < 8; ++k) dup += (loc[k] == loc[j]);
    if (dup == 4) {
      loc[4] = loc[0];
      for (int j = 5; j < 9; ++j) loc[j] = pad;
    }
  }
  for (int j = 0; j < 9; ++j) out[j] = loc[j];
}

extern "C" int nrpn_sort_vertices_f32(const float *vertices, const uint8_t *mask, const int32_t *num_valid, int32_t *idx,
                                      int64_t bn, int m, nrpn_stream_t stream) {
  NRPN_REQUIRE(bn >= 0 && m >= 9 && m <= 64, "sort_vertices: bad sizes bn=%lld m=%d", (long long)bn, m);
  if (bn == 0) return NRPN_OK;
  NRPN_REQUIRE(vertices && mask && num_valid && idx, "sort_vertices: null pointer");
  hipLaunchKernelGGL(sort_vertices_kernel, dim3((unsigned)cdiv64(bn, 128)), dim3(128), 0, as_stream(stream), vertices, mask,
                     num_valid, idx, bn, m);
  NRPN_LAUNCH_CHECK("sort_vertices");
  return NRPN_OK;
}

// =====================================================================================================================
// IoU: paired and all-pairs
// =====================================================================================================================
__global__ void __launch_bounds__(64) iou_pair_obb_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[7], q[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { p[k] = a[i * 7 + k]; q[k] = b[i * 7 + k]; }
  out[i] = geo::iou3d_obb(p, q);
}

template <int W>
__global__ void __launch_bounds__(64) iou_matrix_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, int64_t n,
                                  int64_t m) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = blockIdx.y;
  if (j >= m) return;
  float p[W], q[W];
#pragma unroll
  for (int k = 0; k < W; ++k) { p[k] = a[i * W + k]; q[k] = b[j * W + k]; }
  out[i * m + j] = geo::iou3d<W>(p, q);
}

extern "C" int nrpn_iou3d_obb_pair_f32(const float *b1, const float *b2, float *iou, int64_t n, nrpn_stream_t stream) {
  NRPN_REQUIRE(n >= 0, "iou pair: n<0");
  if (n == 0) return NRPN_OK;
  NRPN_REQUIRE(b1 && b2 && iou, "iou pair: null pointer");
  hipLaunchKernelGGL(iou_pair_obb_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(64), 0, as_stream(stream), b1, b2, iou, n);
  NRPN_LAUNCH_CHECK("iou_pair_obb");
  return NRPN_OK;
}

extern "C" int nrpn_iou3d_matrix_f32(const float *a, const float *b, float *iou, int64_t n, int64_t m, int box_dim,
                                     nrpn_stream_t stream) {
  NRPN_REQUIRE(box_dim == 6 || box_dim == 7, "iou matrix: box_dim must be 6 or 7 (got %d)", box_dim);
  NRPN_REQUIRE(n >= 0 && m >= 0 && n < 65536, "iou matrix: bad sizes");
  if (n == 0 || m == 0) return NRPN_OK;
  NRPN_REQUIRE(a && b && iou, "iou matrix: null pointer");
  dim3 grid((unsigned)cdiv64(m, 64), (unsigned)n);
  if (box_dim == 6)
    hipLaunchKernelGGL(iou_matrix_kernel<6>, grid, dim3(64), 0, as_stream(stream), a, b, iou, n, m);
  else
    hipLaunchKernelGGL(iou_matrix_kernel<7>, grid, dim3(64), 0, as_stream(stream), a, b, iou, n, m);
  NRPN_LAUNCH_CHECK("iou_matrix");
  return NRPN_OK;
}

// =====================================================================================================================
// Bitmask NMS.  Pass 1: 64x64 tiles of the (upper-triangular) suppression matrix, one lane per row, the 64 column
// boxes staged in LDS.  Pass 2: one workgroup per level walks its rows in blocks of 64; the block's mask rows are
// pulled into LDS with one coalesced burst, then wave 0 resolves the 64 sequential decisions from LDS.
// =====================================================================================================================
constexpr int kMaxNms = 16384;
constexpr int kNmsLevels = 64;

// Rotated pairs are expensive (two sin/cos, 16 edge-edge tests, an 8-pick angular selection sort: ~2k instructions) and almost all pairs
// of a tile are far apart, so a tile is worked in two phases: (1) lane = row: a cheap test per column that PROVES the full IoU would be
// exactly 0 -- equal expression for the z overlap (zov == 0 => inter3 = iou2 * u2 * 0 = 0), or BEV circumcircles separated by a 1e-3
// relative margin (no corner inside the other rectangle, no edge crossing => fewer than 3 vertices => area 0) -- on boxes whose numbers
// are ordinary (finite, positive extents: anything else goes to the full computation, whose NaN / inf results suppress as before);
// (2) the surviving (row, column) pairs are compacted into an LDS list and the 64 lanes take them round-robin, so every lane of the
// wave runs the full IoU on a real candidate.  The decisions are those of the one-lane-per-row loop, bit for bit.
__device__ __forceinline__ bool obb_ordinary(const float *b) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 7; ++k) ok = ok && (fabsf(b[k]) < 1e15f);      // also false for NaN / inf
  return ok && b[3] > 1e-15f && b[4] > 1e-15f && b[5] > 1e-15f;
}

__device__ __forceinline__ bool obb_surely_disjoint(const float *p, const float *q) {
  const float zt1 = p[2] + p[5] * 0.5f, zb1 = p[2] - p[5] * 0.5f;
  const float zt2 = q[2] + q[5] * 0.5f, zb2 = q[2] - q[5] * 0.5f;
  const float zov = fmaxf(fminf(zt1, zt2) - fmaxf(zb1, zb2), 0.f);   // the expression of iou3d_obb
  if (zov == 0.f) return true;
  const float dx = p[0] - q[0], dy = p[1] - q[1];
  const float d = sqrtf(dx * dx + dy * dy);
  const float r = 0.5f * (sqrtf(p[3] * p[3] + p[4] * p[4]) + sqrtf(q[3] * q[3] + q[4] * q[4]));
  return d - r > 1e-3f * (r + fabsf(p[0]) + fabsf(p[1]) + fabsf(q[0]) + fabsf(q[1]) + 1.f);
}

constexpr int kNmsMaskThreads = 256;      // four waves per 64 x 64 tile: a dense tile (coarse level: most pairs overlap) is 64 rounds of
                                          // the full IoU for ONE wave -- the critical path of the whole launch -- and 16 for four

template <int W>
__global__ void __launch_bounds__(kNmsMaskThreads)
nms_mask_kernel(const float *__restrict__ boxes, const int32_t *__restrict__ levels,
                                const int32_t *__restrict__ d_count, int n_max, float thr, unsigned long long *__restrict__ mask,
                                int words) {
  const int n = d_count ? min(*d_count, n_max) : n_max;
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb || rb * 64 >= n || cb * 64 >= n) return;
  __shared__ float cbox[64][W + 1];
  __shared__ float rbox[64][W + 1];
  __shared__ int clev[64], rlev[64];
  __shared__ unsigned char cord[64], rord[64];
  __shared__ unsigned long long bits_sh[64];
  __shared__ unsigned short pairs[64 * 64];
  __shared__ int wave_tot[kNmsMaskThreads / 64];
  const int t = threadIdx.x;
  if (t < 128) {
    const bool is_col = t < 64;
    const int l = t & 63;
    const int g = (is_col ? cb : rb) * 64 + l;
    float b[W];
#pragma unroll
    for (int k = 0; k < W; ++k) b[k] = (g < n) ? boxes[(int64_t)g * W + k] : 0.f;
    const int lev = (levels && g < n) ? levels[g] : 0;
    const unsigned char ord = (W == 7) ? (unsigned char)obb_ordinary(b) : 0;
    if (is_col) {
#pragma unroll
      for (int k = 0; k < W; ++k) cbox[l][k] = b[k];
      clev[l] = lev; cord[l] = ord;
    } else {
#pragma unroll
      for (int k = 0; k < W; ++k) rbox[l][k] = b[k];
      rlev[l] = lev; rord[l] = ord;
      bits_sh[l] = 0ull;
    }
  }
  __syncthreads();
  const int cend = min(64, n - cb * 64);
  // phase 1: lane = (row, quarter of the columns): candidate pairs = same level, column after row, not provably disjoint
  const int rl = t & 63, part = t >> 6;
  const int r = rb * 64 + rl;
  unsigned cand = 0u;
  if (r < n) {
    float me[W];
#pragma unroll
    for (int k = 0; k < W; ++k) me[k] = rbox[rl][k];
    const int mylev = rlev[rl];
    const bool prefilter = (W == 7) && (thr >= 0.f) && rord[rl];
    for (int jj = 0; jj < 16; ++jj) {
      const int j = part * 16 + jj;
      const int col = cb * 64 + j;
      if (j < cend && col > r && clev[j] == mylev) {
        float other[W];
#pragma unroll
        for (int k = 0; k < W; ++k) other[k] = cbox[j][k];
        bool skip = false;
        if (W == 7) skip = prefilter && cord[j] && obb_surely_disjoint(me, other);
        if (!skip) cand |= (1u << jj);
      }
    }
  }
  // phase 2: compaction (inclusive scan inside each wave, wave totals through LDS)
  const int cnt = __popc(cand);
  int incl = cnt;
  const int lane = t & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_tot[t >> 6] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kNmsMaskThreads / 64; ++w) {
    if (w < (t >> 6)) base += wave_tot[w];
    total += wave_tot[w];
  }
  int pos = base + incl - cnt;
  unsigned rest = cand;
  while (rest) {
    const int jj = __ffs((int)rest) - 1;
    rest &= rest - 1;
    pairs[pos++] = (unsigned short)((rl << 6) | (part * 16 + jj));
  }
  __syncthreads();
  // phase 3: the full IoU of the surviving pairs, one pair per lane per round
  for (int q = t; q < total; q += kNmsMaskThreads) {
    const int pr = pairs[q];
    const int a_ = pr >> 6, j = pr & 63;
    float a[W], b[W];
#pragma unroll
    for (int k = 0; k < W; ++k) { a[k] = rbox[a_][k]; b[k] = cbox[j][k]; }
    const float v = geo::iou3d<W>(a, b);
    if (!(v <= thr)) atomicOr(&bits_sh[a_], 1ull << j);
  }
  __syncthreads();
  if (t < 64 && rb * 64 + t < n) mask[(int64_t)(rb * 64 + t) * words + cb] = bits_sh[t];
}

__global__ void __launch_bounds__(256)
nms_scan_kernel(const int32_t *__restrict__ levels, const int32_t *__restrict__ d_count, int n_max,
                const unsigned long long *__restrict__ mask, int words, uint8_t *__restrict__ keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
  __shared__ unsigned long long sh_kept;
  const int n = d_count ? min(*d_count, n_max) : n_max;
  const int lev = blockIdx.x;
  // [s, e): rows whose level == lev (levels is non-decreasing); without levels block 0 owns everything
  int s = 0, e = 0;
  if (!levels) {
    if (lev != 0) return;
    e = n;
  } else {
    int lo = 0, hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (levels[mid] < lev) lo = mid + 1; else hi = mid; }
    s = lo;
    hi = n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (levels[mid] <= lev) lo = mid + 1; else hi = mid; }
    e = lo;
  }
  if (s >= e) return;
  const int w0 = s >> 6, w1 = (e - 1) >> 6, nw = w1 - w0 + 1;
  unsigned long long *removed = lds64;          // [nw]
  unsigned long long *rows = lds64 + nw;        // [64][nw]
  const int tid = threadIdx.x;
  for (int w = tid; w < nw; w += blockDim.x) removed[w] = 0ull;
  __syncthreads();
  for (int b0 = s & ~63; b0 < e; b0 += 64) {
    const int bw = b0 >> 6;  // word index of this row block; only columns >= bw matter
    // 16 lanes per row (128-byte segments), 16 rows per pass; words left of the diagonal are never read again and are not staged
    const int wlo = max(0, bw - w0);
    for (int rr = tid >> 4; rr < 64; rr += (int)blockDim.x >> 4) {
      const int row = b0 + rr;
      const bool live = row >= s && row < e;
      for (int w = wlo + (tid & 15); w < nw; w += 16)
        rows[rr * nw + w] = live ? mask[(int64_t)row * words + (w0 + w)] : 0ull;
    }
    __syncthreads();
    if (tid < 64) {
      // the 64 sequential decisions of the block on its diagonal word, in scalar registers: lane l holds row l's diagonal word,
      // R = the suppression word of these 64 rows so far (uniform), row rr is kept iff its bit in R is still clear
      const int row = b0 + tid;
      const bool valid = row >= s && row < e;
      const unsigned long long D = valid ? rows[tid * nw + (bw - w0)] : 0ull;
      const unsigned long long V = __ballot(valid);
      unsigned long long R = removed[bw - w0];
      unsigned long long K = 0ull;
      const int dlo = (int)(unsigned)D, dhi = (int)(unsigned)(D >> 32);
#pragma unroll
      for (int rr = 0; rr < 64; ++rr) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane(dlo, rr), hi = (unsigned)__builtin_amdgcn_readlane(dhi, rr);
        const bool alive = ((V >> rr) & 1ull) && !((R >> rr) & 1ull);
        if (alive) {
          R |= ((unsigned long long)hi << 32) | lo;
          K |= 1ull << rr;
        }
      }
      if ((K >> tid) & 1ull) keep[row] = 1;
      if (tid == 0) sh_kept = K;
    }
    __syncthreads();
    // the kept rows suppress in the words to the right of the diagonal: one lane per word, independent LDS reads
    const unsigned long long K = sh_kept;
    for (int w = tid; w < nw; w += blockDim.x) {
      if (w0 + w <= bw) continue;
      unsigned long long acc = removed[w], rest = K;
      while (rest) {
        const int rr = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        acc |= rows[rr * nw + w];
      }
      removed[w] = acc;
    }
    __syncthreads();
  }
}

extern "C" size_t nrpn_nms3d_workspace_bytes(int64_t n_max) {
  if (n_max <= 0) return 0;
  const int64_t words = (n_max + 63) / 64;
  return (size_t)(n_max * words * 8);
}

extern "C" int nrpn_nms3d(const float *boxes, const int32_t *levels, const int32_t *d_count, int64_t n_max, int box_dim, float thr,
                          uint8_t *keep, void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(box_dim == 6 || box_dim == 7, "nms3d: box_dim must be 6 or 7 (got %d)", box_dim);
  NRPN_REQUIRE(n_max >= 0 && n_max <= kMaxNms, "nms3d: n_max=%lld outside [0,%d]", (long long)n_max, kMaxNms);
  if (n_max == 0) return NRPN_OK;
  NRPN_REQUIRE(boxes && keep && workspace, "nms3d: null pointer");
  const int words = (int)((n_max + 63) / 64);
  hipStream_t st = as_stream(stream);
  NRPN_HIP(hipMemsetAsync(keep, 0, (size_t)n_max, st));
  dim3 grid(words, words);
  auto *mask = reinterpret_cast<unsigned long long *>(workspace);
  if (box_dim == 6)
    hipLaunchKernelGGL(nms_mask_kernel<6>, grid, dim3(kNmsMaskThreads), 0, st, boxes, levels, d_count, (int)n_max, thr, mask, words);
  else
    hipLaunchKernelGGL(nms_mask_kernel<7>, grid, dim3(kNmsMaskThreads), 0, st, boxes, levels, d_count, (int)n_max, thr, mask, words);
  NRPN_LAUNCH_CHECK("nms_mask");
  const size_t lds = (size_t)words * 65 * 8;  // worst case: one level spans every word
  NRPN_LDS(nms_scan_kernel, (int)((kMaxNms / 64) * 65 * 8));
  hipLaunchKernelGGL(nms_scan_kernel, dim3(levels ? kNmsLevels : 1), dim3(256), lds, st, levels, d_count, (int)n_max, mask, words,
                     keep);
  NRPN_LAUNCH_CHECK("nms_scan");
  return NRPN_OK;
}

// =====================================================================================================================
// Segmented top-k: one 1024-thread workgroup per segment.  3-pass radix select on the order-preserving key (LDS
// histograms), collection of the winners, then an LDS bitonic sort of (key, ~index) so the output order is
// (score desc, index asc) -- deterministic, unlike torch.topk's unspecified tie order.
// =====================================================================================================================
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

constexpr int kTopkThreads = 1024;

// exclusive scan of one value per thread (blockDim == 1024) through LDS scratch[1024]; returns (exclusive, total)
__device__ __forceinline__ int block_excl_scan(int v, int *scratch, int *total) {
  const int t = threadIdx.x;
  scratch[t] = v;
  __syncthreads();
  for (int off = 1; off < kTopkThreads; off <<= 1) {
    int add = (t >= off) ? scratch[t - off] : 0;
    __syncthreads();
    scratch[t] += add;
    __syncthreads();
  }
  const int incl = scratch[t];
  *total = scratch[kTopkThreads - 1];
  __syncthreads();
  return incl - v;
}

__device__ __forceinline__ void bitonic_sort_desc(unsigned long long *a, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = a[i], y = a[ixj];
          const bool desc = ((i & k) == 0);
          if (desc ? (x < y) : (x > y)) { a[i] = y; a[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
}

struct TopkSeg { long long begin, end; };

__global__ void __launch_bounds__(kTopkThreads)
topk_kernel(const float *__restrict__ scores, long long seg_begin, long long seg_end, int k, int P, int32_t *__restrict__ out_idx,
            float *__restrict__ out_val) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
  unsigned long long *items = lds64;                            // [P]
  int *hist = reinterpret_cast<int *>(lds64 + P);               // [2048]
  int *scratch = hist + 2048;                                   // [1024]
  __shared__ int sh_digit, sh_need, sh_cnt;
  const int t = threadIdx.x;
  const long long n = seg_end - seg_begin;
  const float *s = scores + seg_begin;
  const int kk = (int)min((long long)k, n);
  for (int i = t; i < P; i += kTopkThreads) items[i] = 0ull;
  if (t == 0) sh_cnt = 0;
  __syncthreads();
  if (n <= k) {
    for (long long i = t; i < n; i += kTopkThreads)
      items[i] = ((unsigned long long)f2key(s[i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    __syncthreads();
  } else {
    unsigned prefix = 0, pmask = 0;
    int need = k;
    const int shifts[3] = {21, 10, 0};
    const int bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      const int nb = 1 << bits[pass];
      for (int i = t; i < 2048; i += kTopkThreads) hist[i] = 0;
      __syncthreads();
      for (long long i = t; i < n; i += kTopkThreads) {
        const unsigned key = f2key(s[i]);
        if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shifts[pass]) & (nb - 1)], 1);
      }
      __syncthreads();
      // suffix counts: thread t owns bins 2t, 2t+1 (nb <= 2048)
      const int b0 = 2 * t, b1 = 2 * t + 1;
      const int c0 = (b0 < nb) ? hist[b0] : 0, c1 = (b1 < nb) ? hist[b1] : 0;
      int tot;
      const int excl = block_excl_scan(c0 + c1, scratch, &tot);   // elements in bins < 2t
      const int above1 = tot - excl - c0 - c1;                     // elements in bins > b1
      const int above0 = above1 + c1;                              // elements in bins > b0
      if (b1 < nb && above1 < need && need <= above1 + c1) { sh_digit = b1; sh_need = need - above1; }
      if (b0 < nb && above0 < need && need <= above0 + c0) { sh_digit = b0; sh_need = need - above0; }
      __syncthreads();
      prefix |= ((unsigned)sh_digit) << shifts[pass];
      pmask |= ((unsigned)(nb - 1)) << shifts[pass];
      need = sh_need;
      __syncthreads();
    }
    const unsigned T = prefix;   // k-th largest key; take every key > T and `need` of the keys == T
    // keys > T
    for (long long i = t; i < n; i += kTopkThreads) {
      const unsigned key = f2key(s[i]);
      if (key > T) {
        const int slot = atomicAdd(&sh_cnt, 1);
        items[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
      }
    }
    __syncthreads();
    const int base = sh_cnt;  // == k - need
    // keys == T: the `need` smallest indices, found with a contiguous-range ordered pass
    const long long chunk = (n + kTopkThreads - 1) / kTopkThreads;
    const long long lo = min(n, (long long)t * chunk), hi = min(n, lo + chunk);
    int mine = 0;
    for (long long i = lo; i < hi; ++i) mine += (f2key(s[i]) == T) ? 1 : 0;
    int tot;
    int rank = block_excl_scan(mine, scratch, &tot);
    for (long long i = lo; i < hi && rank < need; ++i) {
      if (f2key(s[i]) == T) {
        items[base + rank] = ((unsigned long long)T << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        ++rank;
      }
    }
    __syncthreads();
  }
  bitonic_sort_desc(items, P);
  for (int i = t; i < k; i += kTopkThreads) {
    if (i < kk) {
      const unsigned long long it = items[i];
      out_idx[i] = (int32_t)(seg_begin + (long long)(0xFFFFFFFFu - (unsigned)(it & 0xFFFFFFFFull)));
      out_val[i] = key2f((unsigned)(it >> 32));
    } else {
      out_idx[i] = -1;
      out_val[i] = -INFINITY;
    }
  }
}

static int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

extern "C" int nrpn_segmented_topk_f32(const float *scores, const int64_t *h_offsets, int nseg, int k, int32_t *out_idx,
                                       float *out_val, nrpn_stream_t stream) {
  NRPN_REQUIRE(nseg >= 0 && k >= 1 && k <= 16384, "topk: bad nseg=%d k=%d", nseg, k);
  if (nseg == 0) return NRPN_OK;
  NRPN_REQUIRE(scores && h_offsets && out_idx && out_val, "topk: null pointer");
  const int P = next_pow2(k);
  const size_t lds = (size_t)P * 8 + 2048 * 4 + 1024 * 4;
  NRPN_LDS(topk_kernel, 16384 * 8 + 2048 * 4 + 1024 * 4);
  for (int s = 0; s < nseg; ++s) {
    NRPN_REQUIRE(h_offsets[s + 1] >= h_offsets[s] && h_offsets[s + 1] < (1ll << 31), "topk: bad segment %d", s);
    hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(kTopkThreads), lds, as_stream(stream), scores, (long long)h_offsets[s],
                       (long long)h_offsets[s + 1], k, P, out_idx + (int64_t)s * k, out_val + (int64_t)s * k);
  }
  NRPN_LAUNCH_CHECK("topk");
  return NRPN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Long segments (level 0 of a 200 x 200 x 130 scene holds ~1.07 M scores): the single workgroup above walks the segment five
// times (2.2 ms).  Same selection, spread over G workgroups of 256 lanes, each on a contiguous slice:
//   hist x3   LDS histogram of the slice for the current digit -> integer atomics into the segment's global histogram
//             (the digits already fixed are re-derived by every workgroup from the finished histograms: no host round trip);
//   collect   keys above the k-th key T go to the item list through one atomic counter (their order is irrelevant: the
//             final sort orders (key, index)); the slice's count of keys == T is written per workgroup;
//   ties      the `need` smallest indices among the keys == T: slice order = index order, ranks from the per-slice counts;
//   sort      one workgroup: bitonic sort of the k items in LDS, outputs as above.
// Integer atomics only: every run produces the same histogram, the same T, the same set and the same order.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kTopkSliceThreads = 256;
constexpr int kTopkMaxSlices = 256;
constexpr int kTopkLongSegment = 65536;
struct TopkWs {                 // per segment, in the caller's workspace (zeroed by the launcher)
  int hist[3][2048];
  int count_above;              // atomic slot counter of the collect pass
  int pad[3];
  int ties[kTopkMaxSlices];     // keys == T per slice
};

// digit of the need-th largest element of a histogram (bins descending) with 256 lanes; returns through shared memory
__device__ __forceinline__ void topk_select_digit(const int *hist, int nb, int need, int *scratch, int *sh_out) {
  const int t = threadIdx.x;
  const int per = 2048 / kTopkSliceThreads;       // 8 bins per lane
  int c[per], mine = 0;
#pragma unroll
  for (int i = 0; i < per; ++i) { const int b = t * per + i; c[i] = (b < nb) ? hist[b] : 0; mine += c[i]; }
  scratch[t] = mine;
  __syncthreads();
  for (int off = 1; off < kTopkSliceThreads; off <<= 1) {
    const int add = (t >= off) ? scratch[t - off] : 0;
    __syncthreads();
    scratch[t] += add;
    __syncthreads();
  }
  const int total = scratch[kTopkSliceThreads - 1];
  int above = total - scratch[t];               // elements in bins above my eight
  __syncthreads();
#pragma unroll
  for (int i = per - 1; i >= 0; --i) {
    if (above < need && need <= above + c[i]) { sh_out[0] = t * per + i; sh_out[1] = need - above; }
    above += c[i];
  }
  __syncthreads();
}

// (prefix, pmask, need) after `passes` finished histograms
__device__ __forceinline__ void topk_resolve(const TopkWs *ws, int k, int passes, int *scratch, int *sh_out, unsigned *prefix,
                                             unsigned *pmask, int *need) {
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  *prefix = 0; *pmask = 0; *need = k;
  for (int p = 0; p < passes; ++p) {
    const int nb = 1 << bits[p];
    topk_select_digit(ws->hist[p], nb, *need, scratch, sh_out);
    *prefix |= ((unsigned)sh_out[0]) << shifts[p];
    *pmask |= ((unsigned)(nb - 1)) << shifts[p];
    *need = sh_out[1];
    __syncthreads();
  }
}

__global__ void __launch_bounds__(kTopkSliceThreads)
topk_hist_kernel(const float *__restrict__ scores, long long seg_begin, long long seg_end, int k, int pass, TopkWs *__restrict__ ws) {
  __shared__ int hist[2048];
  __shared__ int scratch[kTopkSliceThreads];
  __shared__ int sh_out[2];
  const int t = threadIdx.x;
  const long long n = seg_end - seg_begin;
  const float *s = scores + seg_begin;
  unsigned prefix, pmask;
  int need;
  topk_resolve(ws, k, pass, scratch, sh_out, &prefix, &pmask, &need);
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  const int nb = 1 << bits[pass], shift = shifts[pass];
  for (int i = t; i < 2048; i += kTopkSliceThreads) hist[i] = 0;
  __syncthreads();
  const long long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = min(n, (long long)blockIdx.x * chunk), hi = min(n, lo + chunk);
  for (long long i = lo + t; i < hi; i += kTopkSliceThreads) {
    const unsigned key = f2key(s[i]);
    if ((key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & (nb - 1)], 1);
  }
  __syncthreads();
  for (int i = t; i < nb; i += kTopkSliceThreads)
    if (hist[i]) atomicAdd(&ws->hist[pass][i], hist[i]);
}

__global__ void __launch_bounds__(kTopkSliceThreads)
topk_collect_kernel(const float *__restrict__ scores, long long seg_begin, long long seg_end, int k, TopkWs *__restrict__ ws,
                    unsigned long long *__restrict__ items) {
  __shared__ int scratch[kTopkSliceThreads];
  __shared__ int sh_out[2];
  __shared__ int sh_ties;
  const int t = threadIdx.x;
  const long long n = seg_end - seg_begin;
  const float *s = scores + seg_begin;
  unsigned T, pmask;
  int need;
  topk_resolve(ws, k, 3, scratch, sh_out, &T, &pmask, &need);
  if (t == 0) sh_ties = 0;
  __syncthreads();
  const long long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = min(n, (long long)blockIdx.x * chunk), hi = min(n, lo + chunk);
  int ties = 0;
  for (long long i = lo + t; i < hi; i += kTopkSliceThreads) {
    const unsigned key = f2key(s[i]);
    if (key > T) {
      const int slot = atomicAdd(&ws->count_above, 1);      // k - need of them in the whole segment
      items[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
    ties += (key == T) ? 1 : 0;
  }
  if (ties) atomicAdd(&sh_ties, ties);
  __syncthreads();
  if (t == 0) ws->ties[blockIdx.x] = sh_ties;
}

__global__ void __launch_bounds__(kTopkSliceThreads)
topk_ties_kernel(const float *__restrict__ scores, long long seg_begin, long long seg_end, int k, const TopkWs *__restrict__ ws,
                 unsigned long long *__restrict__ items) {
  __shared__ int scratch[kTopkSliceThreads];
  __shared__ int sh_out[2];
  const int t = threadIdx.x;
  const long long n = seg_end - seg_begin;
  const float *s = scores + seg_begin;
  unsigned T, pmask;
  int need;
  topk_resolve(ws, k, 3, scratch, sh_out, &T, &pmask, &need);
  int before = 0;                                            // ties in the slices in front of mine
  for (int g = t; g < (int)blockIdx.x; g += kTopkSliceThreads) before += ws->ties[g];
  scratch[t] = before;
  __syncthreads();
  for (int off = kTopkSliceThreads / 2; off > 0; off >>= 1) {
    if (t < off) scratch[t] += scratch[t + off];
    __syncthreads();
  }
  before = scratch[0];
  __syncthreads();
  if (before >= need) return;                                // uniform over the workgroup
  const long long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = min(n, (long long)blockIdx.x * chunk), hi = min(n, lo + chunk);
  const long long sub = (hi - lo + kTopkSliceThreads - 1) / kTopkSliceThreads;
  const long long a = min(hi, lo + (long long)t * sub), b = min(hi, a + sub);
  int mine = 0;
  for (long long i = a; i < b; ++i) mine += (f2key(s[i]) == T) ? 1 : 0;
  scratch[t] = mine;
  __syncthreads();
  for (int off = 1; off < kTopkSliceThreads; off <<= 1) {
    const int add = (t >= off) ? scratch[t - off] : 0;
    __syncthreads();
    scratch[t] += add;
    __syncthreads();
  }
  int rank = before + scratch[t] - mine;
  const int base = k - need;
  for (long long i = a; i < b && rank < need; ++i) {
    if (f2key(s[i]) == T) {
      items[base + rank] = ((unsigned long long)T << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
      ++rank;
    }
  }
}

__global__ void __launch_bounds__(kTopkThreads)
topk_sort_kernel(const unsigned long long *__restrict__ items_in, long long seg_begin, int k, int P, int32_t *__restrict__ out_idx,
                 float *__restrict__ out_val) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
  const int t = threadIdx.x;
  for (int i = t; i < P; i += kTopkThreads) lds64[i] = (i < k) ? items_in[i] : 0ull;
  __syncthreads();
  bitonic_sort_desc(lds64, P);
  for (int i = t; i < k; i += kTopkThreads) {
    const unsigned long long it = lds64[i];
    out_idx[i] = (int32_t)(seg_begin + (long long)(0xFFFFFFFFu - (unsigned)(it & 0xFFFFFFFFull)));
    out_val[i] = key2f((unsigned)(it >> 32));
  }
}

extern "C" size_t nrpn_segmented_topk_workspace_bytes(int nseg, int k) {
  if (nseg <= 0 || k <= 0) return 0;
  return (size_t)nseg * (sizeof(TopkWs) + (size_t)next_pow2(k) * 8);
}

extern "C" int nrpn_segmented_topk_f32_ws(const float *scores, const int64_t *h_offsets, int nseg, int k, int32_t *out_idx,
                                          float *out_val, void *workspace, size_t workspace_bytes, nrpn_stream_t stream) {
  NRPN_REQUIRE(nseg >= 0 && k >= 1 && k <= 16384, "topk: bad nseg=%d k=%d", nseg, k);
  if (nseg == 0) return NRPN_OK;
  NRPN_REQUIRE(scores && h_offsets && out_idx && out_val, "topk: null pointer");
  const int P = next_pow2(k);
  bool any_long = false;
  for (int s = 0; s < nseg; ++s) {
    NRPN_REQUIRE(h_offsets[s + 1] >= h_offsets[s] && h_offsets[s + 1] < (1ll << 31), "topk: bad segment %d", s);
    any_long = any_long || (h_offsets[s + 1] - h_offsets[s] >= kTopkLongSegment && h_offsets[s + 1] - h_offsets[s] > k);
  }
  hipStream_t st = as_stream(stream);
  TopkWs *heads = reinterpret_cast<TopkWs *>(workspace);
  unsigned long long *items = reinterpret_cast<unsigned long long *>(heads + nseg);
  if (any_long) {
    NRPN_REQUIRE(workspace && workspace_bytes >= nrpn_segmented_topk_workspace_bytes(nseg, k), "topk: workspace too small");
    NRPN_HIP(hipMemsetAsync(heads, 0, sizeof(TopkWs) * (size_t)nseg, st));
  }
  const size_t lds = (size_t)P * 8 + 2048 * 4 + 1024 * 4;
  NRPN_LDS(topk_kernel, 16384 * 8 + 2048 * 4 + 1024 * 4);
  NRPN_LDS(topk_sort_kernel, 16384 * 8);
  for (int s = 0; s < nseg; ++s) {
    const long long b = h_offsets[s], e = h_offsets[s + 1], n = e - b;
    if (n < kTopkLongSegment || n <= k) {
      hipLaunchKernelGGL(topk_kernel, dim3(1), dim3(kTopkThreads), lds, st, scores, b, e, k, P, out_idx + (int64_t)s * k,
                         out_val + (int64_t)s * k);
      continue;
    }
    const int G = (int)std::min<long long>(kTopkMaxSlices, (n + 4095) / 4096);
    unsigned long long *it = items + (size_t)s * P;
    for (int pass = 0; pass < 3; ++pass)
      hipLaunchKernelGGL(topk_hist_kernel, dim3(G), dim3(kTopkSliceThreads), 0, st, scores, b, e, k, pass, heads + s);
    hipLaunchKernelGGL(topk_collect_kernel, dim3(G), dim3(kTopkSliceThreads), 0, st, scores, b, e, k, heads + s, it);
    hipLaunchKernelGGL(topk_ties_kernel, dim3(G), dim3(kTopkSliceThreads), 0, st, scores, b, e, k, heads + s, it);
    hipLaunchKernelGGL(topk_sort_kernel, dim3(1), dim3(kTopkThreads), (size_t)P * 8, st, it, b, k, P, out_idx + (int64_t)s * k,
                       out_val + (int64_t)s * k);
  }
  NRPN_LAUNCH_CHECK("topk");
  return NRPN_OK;
}

// =====================================================================================================================
// anchors / coders
// =====================================================================================================================
extern "C" int64_t nrpn_anchor_table_words(int levels, int anchors_per_cell) {
  return 2 + 8ll * levels + 6ll * levels * anchors_per_cell;
}

__global__ void anchors_kernel(const int32_t *__restrict__ tab, const int64_t *__restrict__ sel, int64_t count,
                               float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const geo::AnchorCell c = geo::anchor_at(tab, sel ? sel[i] : i);
#pragma unroll
  for (int k = 0; k < 6; ++k) out[i * 6 + k] = c.box[k];
}

template <int CODER>
__global__ void decode_kernel(const int32_t *__restrict__ tab, const float *__restrict__ deltas, const int64_t *__restrict__ sel,
                              int64_t count, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int64_t f = sel ? sel[i] : i;
  constexpr int DW = CODER ? 8 : 6, BW = CODER ? 7 : 6;
  float o[BW];
  if (f < 0) {
#pragma unroll
    for (int k = 0; k < BW; ++k) out[i * BW + k] = 0.f;
    return;
  }
  const geo::AnchorCell c = geo::anchor_at(tab, f);
  float d[DW];
#pragma unroll
  for (int k = 0; k < DW; ++k) d[k] = deltas[f * DW + k];
  if (CODER) geo::decode_midpoint(d, c.box, o);
  else geo::decode_aabb(d, c.box, o);
#pragma unroll
  for (int k = 0; k < BW; ++k) out[i * BW + k] = o[k];
}

template <int CODER>
__global__ void encode_kernel(const int32_t *__restrict__ tab, const float *__restrict__ gt, const int64_t *__restrict__ sel,
                              int64_t count, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  constexpr int DW = CODER ? 8 : 6, BW = CODER ? 7 : 6;
  const geo::AnchorCell c = geo::anchor_at(tab, sel ? sel[i] : i);
  float g[BW], o[DW];
#pragma unroll
  for (int k = 0; k < BW; ++k) g[k] = gt[i * BW + k];
  if (CODER) geo::encode_midpoint(g, c.box, o);
  else geo::encode_aabb(g, c.box, o);
#pragma unroll
  for (int k = 0; k < DW; ++k) out[i * DW + k] = o[k];
}

__global__ void obb_to_aabb_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[7], r[6];
#pragma unroll
  for (int k = 0; k < 7; ++k) o[k] = in[i * 7 + k];
  geo::obb_to_aabb(o, r);
#pragma unroll
  for (int k = 0; k < 6; ++k) out[i * 6 + k] = r[k];
}

extern "C" int nrpn_anchors_f32(const int32_t *table, const int64_t *sel, int64_t count, float *anchors, nrpn_stream_t stream) {
  NRPN_REQUIRE(count >= 0, "anchors: count<0");
  if (count == 0) return NRPN_OK;
  NRPN_REQUIRE(table && anchors, "anchors: null pointer");
  hipLaunchKernelGGL(anchors_kernel, dim3((unsigned)cdiv64(count, 256)), dim3(256), 0, as_stream(stream), table, sel, count, anchors);
  NRPN_LAUNCH_CHECK("anchors");
  return NRPN_OK;
}

extern "C" int nrpn_decode_boxes_f32(const int32_t *table, const float *deltas, const int64_t *sel, int64_t count, int coder,
                                     float *boxes, nrpn_stream_t stream) {
  NRPN_REQUIRE(coder == 0 || coder == 1, "decode: coder must be 0 (AABB) or 1 (midpoint), got %d", coder);
  NRPN_REQUIRE(count >= 0, "decode: count<0");
  if (count == 0) return NRPN_OK;
  NRPN_REQUIRE(table && deltas && boxes, "decode: null pointer");
  dim3 grid((unsigned)cdiv64(count, 256));
  if (coder == 0) hipLaunchKernelGGL(decode_kernel<0>, grid, dim3(256), 0, as_stream(stream), table, deltas, sel, count, boxes);
  else hipLaunchKernelGGL(decode_kernel<1>, grid, dim3(256), 0, as_stream(stream), table, deltas, sel, count, boxes);
  NRPN_LAUNCH_CHECK("decode");
  return NRPN_OK;
}

extern "C" int nrpn_encode_boxes_f32(const int32_t *table, const float *gt, const int64_t *sel, int64_t count, int coder,
                                     float *deltas, nrpn_stream_t stream) {
  NRPN_REQUIRE(coder == 0 || coder == 1, "encode: coder must be 0 (AABB) or 1 (midpoint), got %d", coder);
  NRPN_REQUIRE(count >= 0, "encode: count<0");
  if (count == 0) return NRPN_OK;
  NRPN_REQUIRE(table && gt && deltas, "encode: null pointer");
  dim3 grid((unsigned)cdiv64(count, 256));
  if (coder == 0) hipLaunchKernelGGL(encode_kernel<0>, grid, dim3(256), 0, as_stream(stream), table, gt, sel, count, deltas);
  else hipLaunchKernelGGL(encode_kernel<1>, grid, dim3(256), 0, as_stream(stream), table, gt, sel, count, deltas);
  NRPN_LAUNCH_CHECK("encode");
  return NRPN_OK;
}

template <int CODER, bool ENC>
__global__ void coder_pairs_kernel(const float *__restrict__ a, const float *__restrict__ anchors, int64_t count, float *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  constexpr int DW = CODER ? 8 : 6, BW = CODER ? 7 : 6;
  constexpr int IW = ENC ? BW : DW, OW = ENC ? DW : BW;
  float in[IW], an[6], o[OW];
#pragma unroll
  for (int k = 0; k < IW; ++k) in[k] = a[i * IW + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) an[k] = anchors[i * 6 + k];
  if (ENC) { if (CODER) geo::encode_midpoint(in, an, o); else geo::encode_aabb(in, an, o); }
  else { if (CODER) geo::decode_midpoint(in, an, o); else geo::decode_aabb(in, an, o); }
#pragma unroll
  for (int k = 0; k < OW; ++k) out[i * OW + k] = o[k];
}

extern "C" int nrpn_coder_pairs_f32(const float *in, const float *anchors, int64_t count, int coder, int encode, float *out,
                                    nrpn_stream_t stream) {
  NRPN_REQUIRE(coder == 0 || coder == 1, "coder_pairs: coder must be 0 or 1 (got %d)", coder);
  NRPN_REQUIRE(count >= 0, "coder_pairs: count<0");
  if (count == 0) return NRPN_OK;
  NRPN_REQUIRE(in && anchors && out, "coder_pairs: null pointer");
  dim3 grid((unsigned)cdiv64(count, 256));
  hipStream_t st = as_stream(stream);
  if (coder == 0 && !encode) hipLaunchKernelGGL((coder_pairs_kernel<0, false>), grid, dim3(256), 0, st, in, anchors, count, out);
  else if (coder == 0) hipLaunchKernelGGL((coder_pairs_kernel<0, true>), grid, dim3(256), 0, st, in, anchors, count, out);
  else if (!encode) hipLaunchKernelGGL((coder_pairs_kernel<1, false>), grid, dim3(256), 0, st, in, anchors, count, out);
  else hipLaunchKernelGGL((coder_pairs_kernel<1, true>), grid, dim3(256), 0, st, in, anchors, count, out);
  NRPN_LAUNCH_CHECK("coder_pairs");
  return NRPN_OK;
}

// head output rows [cells][ld] (fp32; columns [0,A) logits, [A, A + A*dw) deltas) <-> flat per-scene logits [T], deltas [T,dw]
template <typename T, bool FWD>
__global__ void head_flatten_kernel(void *__restrict__ head, int64_t cells, int ld, int A, int dw, float *__restrict__ logits,
                                    float *__restrict__ deltas, const float *__restrict__ scale2) {
  const int per = A * (1 + dw);
  const int64_t total = cells * (FWD ? per : ld);
  const float s0 = (!FWD && scale2) ? scale2[0] : 1.f, s1 = (!FWD && scale2) ? scale2[1] : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (FWD) {
      const int64_t cell = i / per;
      const int k = (int)(i - cell * per);
      const float v = reinterpret_cast<const float *>(head)[cell * ld + k];
      if (k < A) logits[cell * A + k] = v;
      else deltas[cell * A * dw + (k - A)] = v;
    } else {
      const int64_t cell = i / ld;
      const int k = (int)(i - cell * ld);
      float v = 0.f;
      if (k < A) v = logits[cell * A + k] * s0;
      else if (k < per) v = deltas[cell * A * dw + (k - A)] * s1;
      elem<T>::st(reinterpret_cast<T *>(head) + i, v);
    }
  }
}

extern "C" int nrpn_head_flatten_f32(const float *head, int64_t cells, int ld, int anchors_per_cell, int dw, float *logits, float *deltas,
                                     nrpn_stream_t stream) {
  NRPN_REQUIRE(head && logits && deltas && cells > 0 && anchors_per_cell * (1 + dw) <= ld, "head_flatten: bad args");
  const int64_t total = cells * anchors_per_cell * (1 + dw);
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL((head_flatten_kernel<float, true>), dim3(blocks), dim3(256), 0, as_stream(stream), (void *)head, cells, ld,
                     anchors_per_cell, dw, logits, deltas, (const float *)nullptr);
  NRPN_LAUNCH_CHECK("head_flatten");
  return NRPN_OK;
}

extern "C" int nrpn_head_unflatten(const float *g_logits, const float *g_deltas, int64_t cells, int ld, int anchors_per_cell, int dw,
                                   const float *scale2, void *d_head, int dtype, nrpn_stream_t stream) {
  NRPN_REQUIRE(d_head && g_logits && g_deltas && cells > 0 && anchors_per_cell * (1 + dw) <= ld, "head_unflatten: bad args");
  const int64_t total = cells * ld;
  int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
  if (dtype == NRPN_F32)
    hipLaunchKernelGGL((head_flatten_kernel<float, false>), dim3(blocks), dim3(256), 0, as_stream(stream), d_head, cells, ld,
                       anchors_per_cell, dw, (float *)g_logits, (float *)g_deltas, scale2);
  else
    hipLaunchKernelGGL((head_flatten_kernel<unsigned short, false>), dim3(blocks), dim3(256), 0, as_stream(stream), d_head, cells, ld,
                       anchors_per_cell, dw, (float *)g_logits, (float *)g_deltas, scale2);
  NRPN_LAUNCH_CHECK("head_unflatten");
  return NRPN_OK;
}

extern "C" int nrpn_obb_to_aabb_f32(const float *obb, float *aabb, int64_t n, nrpn_stream_t stream) {
  NRPN_REQUIRE(n >= 0, "obb_to_aabb: n<0");
  if (n == 0) return NRPN_OK;
  NRPN_REQUIRE(obb && aabb, "obb_to_aabb: null pointer");
  hipLaunchKernelGGL(obb_to_aabb_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, as_stream(stream), obb, aabb, n);
  NRPN_LAUNCH_CHECK("obb_to_aabb");
  return NRPN_OK;
}

// =====================================================================================================================
// Proposal filter: one 1024-thread workgroup, three stable compactions (empty slots; OBB centre test on the boxes
// ONLY -- quirk B3; small-box + score threshold), each a block scan over thread-contiguous runs of 16 entries.
// =====================================================================================================================
constexpr int kFilterMax = 16384;
constexpr int kPerThread = kFilterMax / kTopkThreads;  // 16

template <int W>
__global__ void __launch_bounds__(kTopkThreads)
filter_kernel(const float *__restrict__ boxes, const float *__restrict__ logits, const int32_t *__restrict__ levels,
              const uint8_t *__restrict__ valid, int n, float sx, float sy, float sz, float min_size, float score_thresh, int fix_clip,
              float *__restrict__ tmp_boxes, float *__restrict__ out_boxes, float *__restrict__ out_scores,
              int32_t *__restrict__ out_levels, int32_t *__restrict__ tmp_idx, int32_t *__restrict__ d_count) {
  __shared__ int scratch[kTopkThreads];
  const int t = threadIdx.x;
  const int lo = t * kPerThread;
  // ---- stage 0: drop empty candidate slots; tmp_idx[j] = source row of the j-th live candidate
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) { const int i = lo + q; cnt += (i < n && valid[i]) ? 1 : 0; }
  int n0;
  int pos = block_excl_scan(cnt, scratch, &n0);
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) { const int i = lo + q; if (i < n && valid[i]) tmp_idx[pos++] = i; }
  __syncthreads();
  // ---- stage 1: boxes only.  AABB: clamp.  OBB: drop boxes whose centre is outside; scores/levels keep their slots.
  cnt = 0;
  bool okb[kPerThread];
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const int j = lo + q;
    okb[q] = false;
    if (j < n0) {
      const float *b = boxes + (int64_t)tmp_idx[j] * W;
      if (W == 6) okb[q] = true;
      else okb[q] = (b[0] >= 0.f && b[0] <= sx) && (b[1] >= 0.f && b[1] <= sy) && (b[2] >= 0.f && b[2] <= sz);
      cnt += okb[q] ? 1 : 0;
    }
  }
  int n1;
  pos = block_excl_scan(cnt, scratch, &n1);
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const int j = lo + q;
    if (j < n0 && okb[q]) {
      const float *b = boxes + (int64_t)tmp_idx[j] * W;
      float *o = tmp_boxes + (int64_t)pos * (W + 1);
      if (W == 6) {
        o[0] = fminf(fmaxf(b[0], 0.f), sx); o[1] = fminf(fmaxf(b[1], 0.f), sy); o[2] = fminf(fmaxf(b[2], 0.f), sz);
        o[3] = fminf(fmaxf(b[3], 0.f), sx); o[4] = fminf(fmaxf(b[4], 0.f), sy); o[5] = fminf(fmaxf(b[5], 0.f), sz);
      } else {
#pragma unroll
        for (int k = 0; k < W; ++k) o[k] = b[k];
      }
      // which candidate's score/level rides with this box: its own (fixed behaviour) or slot `pos` (reference quirk B3)
      o[W] = __int_as_float(fix_clip ? j : pos);
      ++pos;
    }
  }
  __syncthreads();
  // ---- stage 2: remove_small_boxes + score threshold on the (box, paired score) rows
  cnt = 0;
  float sc[kPerThread];
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const int j = lo + q;
    okb[q] = false;
    if (j < n1) {
      const float *b = tmp_boxes + (int64_t)j * (W + 1);
      const int src = tmp_idx[__float_as_int(b[W])];
      const float l = logits[src];
      sc[q] = 1.0f / (1.0f + expf(-l));
      bool big;
      if (W == 6) big = (b[3] - b[0] >= min_size) && (b[4] - b[1] >= min_size) && (b[5] - b[2] >= min_size);
      else big = (b[3] >= min_size) && (b[4] >= min_size) && (b[5] >= min_size);
      okb[q] = big && (sc[q] >= score_thresh);
      cnt += okb[q] ? 1 : 0;
    }
  }
  int n2;
  pos = block_excl_scan(cnt, scratch, &n2);
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const int j = lo + q;
    if (j < n1 && okb[q]) {
      const float *b = tmp_boxes + (int64_t)j * (W + 1);
      const int src = tmp_idx[__float_as_int(b[W])];
#pragma unroll
      for (int k = 0; k < W; ++k) out_boxes[(int64_t)pos * W + k] = b[k];
      out_scores[pos] = sc[q];
      out_levels[pos] = levels[src];
      ++pos;
    }
  }
  if (t == 0) *d_count = n2;
}

extern "C" size_t nrpn_filter_workspace_bytes(int64_t n, int box_dim) { return (size_t)(n * (box_dim + 1) * 4 + n * 4); }

extern "C" int nrpn_filter_candidates_f32(const float *boxes, const float *logits, const int32_t *levels, const uint8_t *cand_valid,
                                          int64_t n, int box_dim, const float *h_grid_size3, float min_size, float score_thresh,
                                          int fix_obb_clip, float *out_boxes, float *out_scores, int32_t *out_levels,
                                          int32_t *d_count, void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(box_dim == 6 || box_dim == 7, "filter: box_dim must be 6 or 7 (got %d)", box_dim);
  NRPN_REQUIRE(n >= 0 && n <= kFilterMax, "filter: n=%lld outside [0,%d]", (long long)n, kFilterMax);
  NRPN_REQUIRE(boxes && logits && levels && cand_valid && h_grid_size3 && out_boxes && out_scores && out_levels && d_count &&
                   workspace, "filter: null pointer");
  float *tmp_boxes = reinterpret_cast<float *>(workspace);
  int32_t *tmp_idx = reinterpret_cast<int32_t *>(tmp_boxes + n * (box_dim + 1));
  const float sx = h_grid_size3[0], sy = h_grid_size3[1], sz = h_grid_size3[2];
  if (box_dim == 6)
    hipLaunchKernelGGL(filter_kernel<6>, dim3(1), dim3(kTopkThreads), 0, as_stream(stream), boxes, logits, levels, cand_valid, (int)n,
                       sx, sy, sz, min_size, score_thresh, fix_obb_clip, tmp_boxes, out_boxes, out_scores, out_levels, tmp_idx, d_count);
  else
    hipLaunchKernelGGL(filter_kernel<7>, dim3(1), dim3(kTopkThreads), 0, as_stream(stream), boxes, logits, levels, cand_valid, (int)n,
                       sx, sy, sz, min_size, score_thresh, fix_obb_clip, tmp_boxes, out_boxes, out_scores, out_levels, tmp_idx, d_count);
  NRPN_LAUNCH_CHECK("filter");
  return NRPN_OK;
}

// final gather of the NMS survivors in (score desc, index asc) order
__global__ void __launch_bounds__(kTopkThreads)
select_kept_kernel(const float *__restrict__ boxes, const float *__restrict__ scores, const int32_t *__restrict__ levels,
                   const uint8_t *__restrict__ keep, const int32_t *__restrict__ d_count, int n_max, int W, int P, int post,
                   float *__restrict__ out_boxes, float *__restrict__ out_scores, float *__restrict__ out_levels,
                   int32_t *__restrict__ d_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds64[];
  __shared__ int kept;
  const int n = d_count ? min(*d_count, n_max) : n_max;
  if (threadIdx.x == 0) kept = 0;
  __syncthreads();
  // the survivors are packed to the front first (slot order is irrelevant: the sort orders (key, index)), so that the bitonic network
  // runs on next_pow2(kept) items instead of next_pow2(candidates)
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (keep[i]) {
      const int slot = atomicAdd(&kept, 1);
      lds64[slot] = ((unsigned long long)f2key(scores[i]) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
    }
  }
  __syncthreads();
  int P2 = 2;
  while (P2 < kept) P2 <<= 1;
  if (P2 > P) P2 = P;
  for (int i = kept + threadIdx.x; i < P2; i += blockDim.x) lds64[i] = 0ull;
  __syncthreads();
  bitonic_sort_desc(lds64, P2);
  const int m = min(kept, post);
  for (int i = threadIdx.x; i < post; i += blockDim.x) {
    if (i < m) {
      const int src = (int)(0xFFFFFFFFu - (unsigned)(lds64[i] & 0xFFFFFFFFull));
      for (int k = 0; k < W; ++k) out_boxes[(int64_t)i * W + k] = boxes[(int64_t)src * W + k];
      out_scores[i] = scores[src];
      out_levels[i] = (float)levels[src];
    } else {
      for (int k = 0; k < W; ++k) out_boxes[(int64_t)i * W + k] = 0.f;
      out_scores[i] = 0.f;
      out_levels[i] = -1.f;
    }
  }
  if (threadIdx.x == 0) *d_out = m;
}

extern "C" int nrpn_select_kept_f32(const float *boxes, const float *scores, const int32_t *levels, const uint8_t *keep,
                                    const int32_t *d_count, int64_t n, int box_dim, int post_top_n, float *out_boxes,
                                    float *out_scores, float *out_levels, int32_t *d_out_count, nrpn_stream_t stream) {
  NRPN_REQUIRE(box_dim == 6 || box_dim == 7, "select: box_dim must be 6 or 7 (got %d)", box_dim);
  NRPN_REQUIRE(n >= 1 && n <= 16384 && post_top_n >= 1, "select: bad n=%lld post=%d", (long long)n, post_top_n);
  NRPN_REQUIRE(boxes && scores && levels && keep && out_boxes && out_scores && out_levels && d_out_count, "select: null pointer");
  const int P = next_pow2((int)n);
  NRPN_LDS(select_kept_kernel, 16384 * 8);
  hipLaunchKernelGGL(select_kept_kernel, dim3(1), dim3(kTopkThreads), (size_t)P * 8, as_stream(stream), boxes, scores, levels, keep,
                     d_count, (int)n, box_dim, P, post_top_n, out_boxes, out_scores, out_levels, d_out_count);
  NRPN_LAUNCH_CHECK("select_kept");
  return NRPN_OK;
}

// =====================================================================================================================
// Matcher: anchors are recomputed from their flat index (never stored); two passes because the "low quality" rule
// needs every ground-truth box's maximum over all anchors first.  Both passes run the identical IoU code so the
// float == comparison of the reference (utils.py:191-211) is exact.
// =====================================================================================================================
constexpr int kMaxGt = 1024;

__device__ __forceinline__ bool anchor_in_padding(const int32_t *tab, const geo::AnchorCell &c, float ox, float oy, float oz) {
  const int32_t *t = tab + 2 + 8 * c.level;
  const int lx = (int)ceilf(ox / (float)t[3]), ly = (int)ceilf(oy / (float)t[4]), lz = (int)ceilf(oz / (float)t[5]);
  return !(c.ix < lx && c.iy < ly && c.iz < lz);
}

template <int PASS>
__global__ void match_kernel(const int32_t *__restrict__ tab, int64_t total, const float *__restrict__ gt, int G, float fg, float bg,
                             int has_ori, float ox, float oy, float oz, float *__restrict__ gtmax, float *__restrict__ labels,
                             int32_t *__restrict__ matched) {
  __shared__ float sgt[kMaxGt * 6];
  __shared__ float smax[kMaxGt];
  for (int i = threadIdx.x; i < G * 6; i += blockDim.x) sgt[i] = gt[i];
  if (PASS == 1)
    for (int i = threadIdx.x; i < G; i += blockDim.x) smax[i] = gtmax[i];
  __syncthreads();
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = f < total;
  const geo::AnchorCell c = geo::anchor_at(tab, live ? f : 0);
  const bool padded = has_ori && anchor_in_padding(tab, c, ox, oy, oz);
  float best = -INFINITY;
  int besti = 0;
  bool attains = false;
  if (PASS == 0) {
    // per-GT maximum over all anchors: wave shuffle max -> LDS -> one global atomic per GT per block
    // (every IoU is >= 0 or exactly -1, so the signed-int view orders correctly)
    for (int i = threadIdx.x; i < G; i += blockDim.x) smax[i] = __int_as_float(0x80808080);
    __syncthreads();
    for (int g = 0; g < G; ++g) {
      float q = live ? (padded ? -1.0f : geo::iou3d_aabb(sgt + g * 6, c.box)) : __int_as_float(0x80808080);
      int qi = __float_as_int(q);
      for (int off = 32; off > 0; off >>= 1) qi = max(qi, __shfl_down(qi, off, 64));
      if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int *>(smax) + g, qi);
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += blockDim.x) atomicMax(reinterpret_cast<int *>(gtmax) + g, __float_as_int(smax[g]));
    return;
  }
  if (!live) return;
  for (int g = 0; g < G; ++g) {
    const float q = padded ? -1.0f : geo::iou3d_aabb(sgt + g * 6, c.box);
    if (q > best) { best = q; besti = g; }
    attains = attains || (q == smax[g]);
  }
  {
    int idx = besti;
    if (best < bg) idx = -1;
    else if (best < fg) idx = -2;
    if (attains) idx = besti;
    float lab = (idx >= 0) ? 1.0f : (idx == -1 ? 0.0f : -1.0f);
    if (padded) lab = -1.0f;
    labels[f] = lab;
    matched[f] = idx < 0 ? 0 : idx;
  }
}

extern "C" int nrpn_match_anchors_f32(const int32_t *table, int64_t total_anchors, const float *gt_aabb, int num_gt, float fg_thresh,
                                      float bg_thresh, const float *h_ori_size3, float *labels, int32_t *matched, float *workspace,
                                      nrpn_stream_t stream) {
  NRPN_REQUIRE(total_anchors > 0, "match: no anchors");
  NRPN_REQUIRE(num_gt >= 1 && num_gt <= kMaxGt, "match: num_gt=%d outside [1,%d] (empty targets are handled by the caller)", num_gt, kMaxGt);
  NRPN_REQUIRE(table && gt_aabb && labels && matched && workspace, "match: null pointer");
  hipStream_t st = as_stream(stream);
  // atomicMax on the int view: every IoU is >= 0 or exactly -1, both order correctly as signed ints; 0x80.. = lowest
  NRPN_HIP(hipMemsetAsync(workspace, 0x80, (size_t)num_gt * 4, st));
  const float ox = h_ori_size3 ? h_ori_size3[0] : 0.f, oy = h_ori_size3 ? h_ori_size3[1] : 0.f, oz = h_ori_size3 ? h_ori_size3[2] : 0.f;
  dim3 grid((unsigned)cdiv64(total_anchors, 256));
  hipLaunchKernelGGL(match_kernel<0>, grid, dim3(256), 0, st, table, total_anchors, gt_aabb, num_gt, fg_thresh, bg_thresh,
                     h_ori_size3 ? 1 : 0, ox, oy, oz, workspace, labels, matched);
  hipLaunchKernelGGL(match_kernel<1>, grid, dim3(256), 0, st, table, total_anchors, gt_aabb, num_gt, fg_thresh, bg_thresh,
                     h_ori_size3 ? 1 : 0, ox, oy, oz, workspace, labels, matched);
  NRPN_LAUNCH_CHECK("match");
  return NRPN_OK;
}

// =====================================================================================================================
// Balanced positive / negative sampler (reference BalancedPositiveNegativeSampler, model/utils.py:35-98: positives = labels >= 1,
// negatives = labels == 0, num_pos = min(#pos, batch * fraction), num_neg = min(#neg, batch - num_pos), a uniformly random subset of
// each via torch.randperm).  The reference's formulation costs two host read-backs (torch.where sizes) and a device sort of ~10^6
// random keys per scene; here every anchor gets a 32-bit key = splitmix64(seed, index) and the k smallest (key, index) of a class ARE
// a uniformly random k-subset: a 4096-bin histogram of the keys' top 12 bits finds the bin the k-th smallest falls into, everything
// below that bin is taken, the bin itself (~count / 4096 candidates) is sorted in LDS for the remainder, and the selected indices are
// sorted ascending (what torch.where order gives the reference's masks).  Integer atomics only: the result is a function of
// (labels, seed), independent of scheduling.  Outputs: pos [max_pos], neg [batch] int64 ascending, counts = {num_pos, num_neg, error}.
// =====================================================================================================================
constexpr int kSampBins = 4096, kSampCap = 8192;
// workspace int32: hist[2][kSampBins] | cursor[2] bcursor[2] thr[2] need[2] k[2] below[2] err pad[3] | (8-byte aligned) boundary u64 [2][kSampCap]
constexpr int kSampParams = 2 * kSampBins, kSampInts = 2 * kSampBins + 16;

__host__ __device__ __forceinline__ uint32_t sample_key(uint64_t seed, uint64_t i) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}
constexpr uint64_t kSampNegSeed = 0xD1B54A32D192ED03ull;

__device__ __forceinline__ int sample_class(float lab) { return lab >= 1.f ? 0 : (lab == 0.f ? 1 : -1); }

__global__ void __launch_bounds__(256) sample_hist_kernel(const float *__restrict__ labels, int64_t total, uint64_t seed, int *__restrict__ ws) {
  __shared__ int h[2 * kSampBins];
  for (int b = threadIdx.x; b < 2 * kSampBins; b += 256) h[b] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = sample_class(labels[i]);
    if (c < 0) continue;
    const uint32_t key = sample_key(c ? seed ^ kSampNegSeed : seed, (uint64_t)i);
    atomicAdd(&h[c * kSampBins + (key >> 20)], 1);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < 2 * kSampBins; b += 256)
    if (h[b]) atomicAdd(&ws[b], h[b]);
}

__global__ void __launch_bounds__(1024) sample_threshold_kernel(int *__restrict__ ws, int max_pos, int batch) {
  __shared__ int scan[1024];
  __shared__ int kk[2];
  int *par = ws + kSampParams;
  for (int c = 0; c < 2; ++c) {
    const int *h = ws + c * kSampBins;
    const int b0 = threadIdx.x * 4;
    const int v0 = h[b0], v1 = h[b0 + 1], v2 = h[b0 + 2], v3 = h[b0 + 3];
    scan[threadIdx.x] = v0 + v1 + v2 + v3;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive Hillis-Steele scan over the 1024 four-bin groups
      const int add = (int)threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
      __syncthreads();
      scan[threadIdx.x] += add;
      __syncthreads();
    }
    const int total = scan[1023];
    if (threadIdx.x == 0) {
      kk[c] = c == 0 ? min(total, max_pos) : min(total, batch - kk[0]);
      par[8 + c] = kk[c];
      if (kk[c] == 0) { par[4 + c] = -1; par[6 + c] = 0; par[10 + c] = 0; }
      if (kk[c] == total && total > 0) { par[4 + c] = kSampBins; par[6 + c] = 0; par[10 + c] = total; }
    }
    __syncthreads();
    const int k = kk[c];
    if (k > 0 && k < total) {
      int cum = scan[threadIdx.x] - (v0 + v1 + v2 + v3);       // exclusive prefix of this thread's first bin
      const int vs[4] = {v0, v1, v2, v3};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (cum < k && cum + vs[j] >= k) { par[4 + c] = b0 + j; par[6 + c] = k - cum; par[10 + c] = cum; }
        cum += vs[j];
      }
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) sample_collect_kernel(const float *__restrict__ labels, int64_t total, uint64_t seed, int *__restrict__ ws,
                                                             unsigned long long *__restrict__ boundary, int64_t *__restrict__ out_pos,
                                                             int64_t *__restrict__ out_neg) {
  int *par = ws + kSampParams;
  const int thr0 = par[4], thr1 = par[5];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = sample_class(labels[i]);
    if (c < 0) continue;
    const uint32_t key = sample_key(c ? seed ^ kSampNegSeed : seed, (uint64_t)i);
    const int bin = (int)(key >> 20), thr = c ? thr1 : thr0;
    if (bin < thr) {
      const int slot = atomicAdd(&par[c], 1);
      (c ? out_neg : out_pos)[slot] = i;
    } else if (bin == thr) {
      const int slot = atomicAdd(&par[2 + c], 1);
      if (slot < kSampCap) boundary[(size_t)c * kSampCap + slot] = ((unsigned long long)key << 32) | (unsigned long long)i;
    }
  }
}

__device__ void bitonic_sort_u64(unsigned long long *a, int n2) {       // n2 = power of two, all threads of the workgroup participate
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < n2; t += blockDim.x) {
        const int x = t ^ j;
        if (x > t) {
          const unsigned long long u = a[t], v = a[x];
          if (((t & k) == 0) ? (u > v) : (u < v)) { a[t] = v; a[x] = u; }
        }
      }
      __syncthreads();
    }
}

__global__ void __launch_bounds__(1024) sample_finish_kernel(int *__restrict__ ws, const unsigned long long *__restrict__ boundary,
                                                             int64_t *__restrict__ out_pos, int64_t *__restrict__ out_neg, int32_t *__restrict__ counts) {
  __shared__ unsigned long long a[kSampCap];
  const int c = blockIdx.x;
  int *par = ws + kSampParams;
  int64_t *out = c ? out_neg : out_pos;
  const int k = par[8 + c], need = par[6 + c], below = par[10 + c];
  const int nb = par[2 + c];
  if (nb > kSampCap || (need > 0 && nb < need) || par[c] != below) {      // more candidates in one bin than the LDS sort holds / inconsistent counts
    if (threadIdx.x == 0) counts[2] = 1;
    return;
  }
  if (need > 0) {
    int n2 = 1;
    while (n2 < nb) n2 <<= 1;
    for (int t = threadIdx.x; t < n2; t += blockDim.x) a[t] = t < nb ? boundary[(size_t)c * kSampCap + t] : ~0ull;
    __syncthreads();
    bitonic_sort_u64(a, n2);
    for (int t = threadIdx.x; t < need; t += blockDim.x) out[below + t] = (int64_t)(a[t] & 0xffffffffull);
    __syncthreads();
  }
  __threadfence_block();
  int n2 = 1;
  while (n2 < k) n2 <<= 1;
  for (int t = threadIdx.x; t < n2; t += blockDim.x) a[t] = t < k ? (unsigned long long)out[t] : ~0ull;
  __syncthreads();
  bitonic_sort_u64(a, n2);
  for (int t = threadIdx.x; t < k; t += blockDim.x) out[t] = (int64_t)a[t];
  if (threadIdx.x == 0) counts[c] = k;
}

extern "C" size_t nrpn_sample_workspace_bytes(void) { return (size_t)kSampInts * 4 + (size_t)2 * kSampCap * 8; }

extern "C" int nrpn_sample_pos_neg(const float *labels, int64_t total, int max_pos, int batch, int64_t seed, void *workspace, int64_t *out_pos,
                                   int64_t *out_neg, int32_t *counts, nrpn_stream_t stream) {
  NRPN_REQUIRE(total > 0 && total < (1ll << 31), "sample: total=%lld outside (0, 2^31)", (long long)total);
  NRPN_REQUIRE(max_pos >= 0 && batch >= max_pos && batch <= kSampCap, "sample: need 0 <= max_pos <= batch <= %d (got %d, %d)", kSampCap, max_pos, batch);
  NRPN_REQUIRE(labels && workspace && out_pos && out_neg && counts, "sample: null pointer");
  hipStream_t st = as_stream(stream);
  int *ws = reinterpret_cast<int *>(workspace);
  unsigned long long *boundary = reinterpret_cast<unsigned long long *>(ws + kSampInts);
  NRPN_HIP(hipMemsetAsync(ws, 0, (size_t)kSampInts * 4, st));
  NRPN_HIP(hipMemsetAsync(counts, 0, 3 * 4, st));
  const unsigned blocks = (unsigned)std::min<int64_t>(cdiv64(total, 256 * 8), 2048);
  const uint64_t useed = (uint64_t)seed;
  hipLaunchKernelGGL(sample_hist_kernel, dim3(blocks), dim3(256), 0, st, labels, total, useed, ws);
  hipLaunchKernelGGL(sample_threshold_kernel, dim3(1), dim3(1024), 0, st, ws, max_pos, batch);
  hipLaunchKernelGGL(sample_collect_kernel, dim3(blocks), dim3(256), 0, st, labels, total, useed, ws, boundary, out_pos, out_neg);
  hipLaunchKernelGGL(sample_finish_kernel, dim3(2), dim3(1024), 0, st, ws, boundary, out_pos, out_neg, counts);
  NRPN_LAUNCH_CHECK("sample_pos_neg");
  return NRPN_OK;
}

// =====================================================================================================================
// Sampled losses (<= a few hundred rows): one workgroup, wave shuffles + LDS for the two reductions, gradients
// written straight into the (caller-zeroed) dense gradient buffers of the head outputs.
// =====================================================================================================================
__device__ __forceinline__ float block_sum_256(float v, float *sh) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  __syncthreads();
  return r;
}

__global__ void __launch_bounds__(256)
sampled_loss_kernel(const float *__restrict__ logits, const float *__restrict__ deltas, int dw, const float *__restrict__ targets,
                    const int64_t *__restrict__ pos, int64_t npos, const int64_t *__restrict__ neg, int64_t nneg, float beta,
                    float *__restrict__ loss2, float *__restrict__ g_logits, float *__restrict__ g_deltas) {
  __shared__ float sh[4];
  const int64_t ns = npos + nneg;
  const float inv = 1.0f / (float)ns;
  float bce = 0.f, reg = 0.f;
  for (int64_t i = threadIdx.x; i < ns; i += blockDim.x) {
    const bool is_pos = i < npos;
    const int64_t r = is_pos ? pos[i] : neg[i - npos];
    const float x = logits[r], y = is_pos ? 1.f : 0.f;
    bce += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    if (g_logits) g_logits[r] = (1.0f / (1.0f + expf(-x)) - y) * inv;
  }
  for (int64_t i = threadIdx.x; i < npos * dw; i += blockDim.x) {
    const int64_t p = i / dw;
    const int k = (int)(i - p * dw);
    const int64_t r = pos[p];
    const float d = deltas[r * dw + k] - targets[p * dw + k];
    const float ad = fabsf(d);
    reg += (ad < beta) ? 0.5f * d * d / beta : ad - 0.5f * beta;
    if (g_deltas) g_deltas[r * dw + k] = ((ad < beta) ? d / beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * inv;
  }
  const float b = block_sum_256(bce, sh);
  const float g = block_sum_256(reg, sh);
  if (threadIdx.x == 0) { loss2[0] = b * inv; loss2[1] = g * inv; }
}

extern "C" int nrpn_rpn_sampled_loss_f32(const float *logits, const float *deltas, int dw, const float *targets, const int64_t *pos,
                                         int64_t npos, const int64_t *neg, int64_t nneg, float beta, float *loss2, float *g_logits,
                                         float *g_deltas, nrpn_stream_t stream) {
  NRPN_REQUIRE(dw == 6 || dw == 8, "sampled loss: dw must be 6 or 8 (got %d)", dw);
  NRPN_REQUIRE(npos >= 0 && nneg >= 0 && npos + nneg > 0, "sampled loss: empty sample");
  NRPN_REQUIRE(logits && deltas && loss2 && (npos == 0 || (pos && targets)) && (nneg == 0 || neg), "sampled loss: null pointer");
  hipLaunchKernelGGL(sampled_loss_kernel, dim3(1), dim3(256), 0, as_stream(stream), logits, deltas, dw, targets, pos, npos, neg, nneg,
                     beta, loss2, g_logits, g_deltas);
  NRPN_LAUNCH_CHECK("sampled_loss");
  return NRPN_OK;
}

// =====================================================================================================================
// Proposal metrics on the device (reference eval.py:14-81 recall, 319-395 VOC AP)  [a25 / f1]
// =====================================================================================================================
// Greedy GT <-> proposal matching of evaluate_box_proposals_recall: min(P, G) rounds of "take the GT whose best remaining proposal
// overlaps most, record that overlap, retire both".  One workgroup per scene; per-GT (max, arg-max) live in LDS and only the columns
// whose arg-max row has just been retired are rescanned.  torch.max semantics on ties: the FIRST (lowest-index) maximum, for rows and
// for GTs -- a GT that overlaps nothing still retires proposal 0 (or the first remaining one), exactly as the reference does.
constexpr int kRecallMaxGt = 1024;
__global__ void __launch_bounds__(256) recall_match_kernel(float *__restrict__ ov, int P, int G, float *__restrict__ covered) {
  __shared__ float cmax[kRecallMaxGt];
  __shared__ int carg[kRecallMaxGt];
  __shared__ int pick[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto scan_col = [&](int g) {        // one wave: (max, lowest arg-max) of column g over the rows
    float best = -3.0e38f;
    int arg = 0x7fffffff;
    for (int r = lane; r < P; r += 64) {
      const float v = ov[(long long)r * G + g];
      if (v > best) { best = v; arg = r; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oa = __shfl_xor(arg, off, 64);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) { cmax[g] = best; carg[g] = arg; }
  };
  for (int g = wave; g < G; g += 4) scan_col(g);
  __syncthreads();
  const int rounds = min(P, G);
  for (int j = 0; j < rounds; ++j) {
    if (wave == 0) {                  // first GT attaining the largest column maximum
      float best = -3.0e38f;
      int arg = 0x7fffffff;
      for (int g = lane; g < G; g += 64)
        if (cmax[g] > best) { best = cmax[g]; arg = g; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oa = __shfl_xor(arg, off, 64);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
      }
      if (lane == 0) { pick[0] = arg; pick[1] = carg[arg]; covered[j] = best; }
    }
    __syncthreads();
    const int gi = pick[0], bi = pick[1];
    for (int g = tid; g < G; g += 256) ov[(long long)bi * G + g] = -1.f;     // retire the proposal ...
    for (int r = tid; r < P; r += 256) ov[(long long)r * G + gi] = -1.f;     // ... and the ground-truth box
    __syncthreads();
    for (int g = wave; g < G; g += 4)
      if (g == gi || carg[g] == bi) scan_col(g);                              // only these columns changed their maximum
    __syncthreads();
  }
}

extern "C" int nrpn_recall_match_f32(float *overlaps, int num_proposals, int num_gt, float *covered, nrpn_stream_t stream) {
  NRPN_REQUIRE(overlaps && covered && num_proposals > 0 && num_gt > 0, "recall_match: bad args");
  NRPN_REQUIRE(num_gt <= kRecallMaxGt, "recall_match: at most %d ground-truth boxes per scene (got %d)", kRecallMaxGt, num_gt);
  hipLaunchKernelGGL(recall_match_kernel, dim3(1), dim3(256), 0, as_stream(stream), overlaps, num_proposals, num_gt, covered);
  NRPN_LAUNCH_CHECK("recall_match");
  return NRPN_OK;
}

// VOC AP bookkeeping: detection `order[r]` (rank r in the global score order) is a true positive iff its best IoU exceeds the threshold
// and no higher-ranked detection claimed the same (scene, GT) key -- i.e. it holds the MINIMUM rank among the qualifying detections of its
// key.  Pass 1: integer atomicMin of the rank into first[key] (associative: deterministic); pass 2: compare.  first[] must be
// pre-filled with INT_MAX by the caller (ap_mark does it).
__global__ void ap_first_kernel(const int64_t *__restrict__ order, const float *__restrict__ best_iou, const int64_t *__restrict__ key, int64_t n,
                                float thr, int *__restrict__ first) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const long long d = order[r];
  if (best_iou[d] > thr) atomicMin(first + key[d], (int)r);
}
__global__ void ap_tp_kernel(const int64_t *__restrict__ order, const float *__restrict__ best_iou, const int64_t *__restrict__ key, int64_t n,
                             float thr, const int *__restrict__ first, uint8_t *__restrict__ tp) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const long long d = order[r];
  tp[r] = (best_iou[d] > thr && first[key[d]] == (int)r) ? 1 : 0;
}

extern "C" int nrpn_ap_mark(const int64_t *order, const float *best_iou, const int64_t *key, int64_t n, int64_t num_keys, float iou_thresh,
                            int32_t *first_ws, uint8_t *tp, nrpn_stream_t stream) {
  NRPN_REQUIRE(order && best_iou && key && first_ws && tp && n > 0 && num_keys > 0 && n < (1ll << 31), "ap_mark: bad args");
  hipStream_t st = as_stream(stream);
  NRPN_HIP(hipMemsetAsync(first_ws, 0x7f, (size_t)num_keys * 4, st));       // 0x7f7f7f7f > any rank
  const dim3 grid((unsigned)cdiv64(n, 256));
  hipLaunchKernelGGL(ap_first_kernel, grid, dim3(256), 0, st, order, best_iou, key, n, iou_thresh, first_ws);
  hipLaunchKernelGGL(ap_tp_kernel, grid, dim3(256), 0, st, order, best_iou, key, n, iou_thresh, first_ws, tp);
  NRPN_LAUNCH_CHECK("ap_mark");
  return NRPN_OK;
}

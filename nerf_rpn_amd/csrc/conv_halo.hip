// "Halo" form of the 3x3x3 implicit GEMM for the wide bf16 layers of the RPN path (reference feature_extractor.py:345-358, fpn.py:109-110,
// anchor.py:190-198 -- the 256 -> 256 @ 40^3 layers are 66 % of the model's FLOPs).  Own translation unit: it is the kernel under work.
#include "conv_common.cuh"

#include <type_traits>

// ---------------------------------------------------------------------------------------------------------------------
// "Halo" form of the 3x3x3 implicit GEMM for the wide bf16 layers (256 x 256 tile, 8 waves as conv_igemm_big_kernel).
//
// The kernels above stage one 256-row A tile per (tap, channel chunk): 27 shifted copies of almost the same voxels go through the
// L2 -> LDS path, and half of all LDS-DMA pieces (whose issue cost is what stalls the MFMA stream) carry them.  Here a workgroup owns
// a 4 x 8 x 8 BLOCK of output voxels and stages the 6 x 10 x 10 input halo of that block ONCE per 32-channel chunk; all 27 taps read
// their A fragments from it -- a tap is a constant row offset.  Per 13.5 K-steps of 32 MFMAs a wave then issues 5 halo pieces + 56
// weight pieces instead of 216 (-72 % A traffic, -44 % LDS-DMA instructions), and the per-row 27-bit tap masks disappear: halo rows
// outside the grid are zero-filled by the buffer descriptor when they are loaded.
//
// LDS layout of the halo: SLOT-major, A[q][row] with q = 16-byte channel slot (4 per 32-channel chunk) and row = hx * 104 + hy * 10 + hz
// (x-plane pitch 104 = 100 + 4 pad rows).  A fragment read of tap (dx, dy, dz), slot q is then
//     base(lane) + q * R * 16 + ((dx+1) * 104 + (dy+1) * 10 + (dz+1)) * 16        -- ONE per-lane base per 32-row block + an immediate,
// (the XOR-swizzled row-major tiles of the other kernels need a different address per sub-step).  Bank conflicts: the 16 lanes of a
// ds_read_b128 group must hit 16 different 16-byte columns, i.e. rows that differ mod 16.  A 32-row MFMA block is 2x x 2y x 8z voxels;
// with pitches 10 (y) and 104 = 8 mod 16 (x) the rows of {y0: x0 z0..7, x1 z0..7} are 0..7, 8..15 and those of {y1} are 10..17, 18..25
// (mod 16: all distinct), so lane group A = {0-3, 12-15, 20-27} takes the y0 voxels and group B = {4-11, 16-19, 28-31} the y1 voxels;
// a tap offset shifts every lane alike.  B (weights): 2 taps x 32 channels per K-step = the same 256 x 128-byte swizzled tile as before,
// read straight from the [taps][Cout][Cin] operand (two 64-byte runs per row, coalesced LDS-DMA).  K order: chunk outer, tap pair inner (14 K-steps per
// chunk, the last one half empty).  145 KB of LDS.  Used for grids that the 4 x 8 x 8 blocks cover with little waste (40^3: exact).
// ---------------------------------------------------------------------------------------------------------------------
namespace hk {
constexpr int TX = 4, TY = 8, TZ = 8, PY = 10, PX = 104, R = 6 * PX;             // R = 624 halo rows
constexpr int RP = 640;                                                         // rows per slot plane (10 LDS-DMA instructions of 64 rows)
constexpr int A_BYTES = 4 * RP * 16;                                            // 40960: one 32-channel chunk
constexpr int B_BYTES = 256 * 128;
constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;                            // 147456
static_assert(A_BYTES + 2 * RP * 16 + (2 * PX + 2 * PY + 2) * 16 + 16 <= 65536, "A fragment offsets must fit the 16-bit ds immediate");
constexpr int KSTEPS = 14;                                                      // tap pairs per chunk (27 taps)
// row r (0..31) of an MFMA block -> voxel inside the 2 x 2 x 8 block, chosen so that the two ds_read_b128 lane groups are conflict-free
__device__ __forceinline__ void block_voxel(int r, int &xl, int &yl, int &zl) {
  const bool inA = (r < 4) | ((r >= 12) & (r < 16)) | ((r >= 20) & (r < 28));
  const int gi = inA ? (r < 4 ? r : (r < 16 ? r - 8 : r - 12)) : (r < 12 ? r - 4 : (r < 20 ? r - 8 : r - 16));
  yl = inA ? 0 : 1; xl = gi >> 3; zl = gi & 7;
}
}  // namespace hk

// V (tools-only A/B variants, selected by the NRPN_CONV_DEBUG_VARIANT bits; the production instantiation is V = 0 -- set below to the measured
// winner): bit 0 = static priority for the second-dispatched waves 4-7 (MI355X guide, "two waves per SIMD" item 4); bit 1 = waves 4-7 issue
// their weight pieces one sub-step later than waves 0-3 (the two waves of a SIMD are then not both in their DMA-issue phase).
// XP (round 5): taps are paired ACROSS chunk boundaries -- the 4 x 27 (tap, chunk) entries of four consecutive chunks form 54 full K-steps
// (entry e = 27 j + tap -> K-step e / 2) instead of 4 x 14 with a half-empty one per chunk: 108 K-steps instead of 112 for Cin = 256, and
// no K-step pays a barrier + a weight tile for 16 MFMAs.  Four chunks per loop iteration because 27 K-steps per chunk pair is odd: the weight
// double buffer (a compile-time immediate in every fragment address) only returns to buffer 0 after 54.  Needs Cin % 128 == 0.
// OF32: fp32 output rows (NRPN_CONV_OUT_F32: the bf16x3 parity mode feeds split bf16 operands and keeps fp32 activations); no mask / statistics.
template <int K, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {      // f(integral_constant<K>) ... f(integral_constant<N-1>): unrolled by construction
  if constexpr (K < N) {
    f(std::integral_constant<int, K>{});
    static_for<K + 1, N>(f);
  }
}

template <int V, bool XP = false, bool OF32 = false>
__global__ void __launch_bounds__(512, 1) conv_halo_kernel(const ConvArgs p) {
  using namespace hk;
  typedef bf16s T;
  constexpr int TM = 4, TN = 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *Ab = lds;                          // [2][4][R][16]
  char *Bb = lds + 2 * A_BYTES;            // [2][256][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int tx = (p.X + TX - 1) / TX, ty = (p.Y + TY - 1) / TY, tz = (p.Z + TZ - 1) / TZ;
  const unsigned ntiles = (p.Cout + 255) / 256;
  const unsigned mtiles = (unsigned)(p.M / ((long long)p.X * p.Y * p.Z)) * tx * ty * tz;      // scenes x blocks
  const unsigned tile = xcd_remap(blockIdx.x, mtiles * ntiles);
  const int n0 = (int)(tile % ntiles) * 256;
  unsigned t = tile / ntiles;
  const int bz = (int)(t % tz); t /= tz;
  const int by = (int)(t % ty); t /= ty;
  const int bx = (int)(t % tx);
  const long long scene = t / tx;
  const int x0 = bx * TX, y0 = by * TY, z0 = bz * TZ;
  const long long vbase = scene * (long long)p.X * p.Y * p.Z;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
  // ---- halo pieces of this wave: instruction k = wave + 8 j (j = 0..4) covers rows [64 (k % 10), +64) of slot k / 10.  The address of a
  // piece is recomputed when it is issued (once per chunk, ~30 VALU against 432 MFMAs): nothing of it lives in registers across K-steps.
  constexpr int A_PER_WAVE = 5;
  auto issue_halo_piece = [&](int buf, int chunk, int j) {
    const int k = wave_u + 8 * j;                          // 0..39, wave-uniform
    const int q = k / 10, kk = k - q * 10;
    int ln = lane;
    asm volatile("" : "+v"(ln));                           // opaque: keeps the address arithmetic HERE (hoisted out of the chunk loop it
                                                           // becomes five more live registers, spilled and reloaded behind a vmcnt(0))
    const int row = kk * 64 + ln;
    const int hx = row / PX, rem = row - hx * PX, hy = rem / PY, hz = rem - hy * PY;
    const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
    const bool ok = row < R && rem < 100 && (unsigned)gx < (unsigned)p.X && (unsigned)gy < (unsigned)p.Y && (unsigned)gz < (unsigned)p.Z;
    const unsigned off = ok ? (unsigned)(((vbase + ((long long)gx * p.Y + gy) * p.Z + gz) * p.Cin) * 2) + (unsigned)(q * 16 + chunk * 64) : kOOB;
    lds_dma16(xr, Ab + buf * A_BYTES + q * (RP * 16) + kk * 1024, off);
  };
  // ---- weight tile of a K-step: the row-major [256 rows][128 B] tile of the other kernels (16-byte slots XOR-swizzled by (row >> 1) & 7),
  // row = [tap 2 ks: 32 channels | tap 2 ks + 1: 32 channels].  Its LDS-DMA is COALESCED -- 8 lanes fetch the two 64-byte runs of one
  // output row -- which a slot-major weight tile is not: 64 lanes x 16 bytes from 64 different 512-byte rows cost four times the L1
  // cycles and made the weight stream alone as long as the K-step's MFMAs (measured: 223 us against 192 us for the 8-wave kernel).
  const int lr = tid >> 3, lsl = tid & 7;
  const int logical = lsl ^ ((lr >> 1) & 7);             // (r >> 1) & 7 is the same for rows lr + 64 i
  const unsigned b_sel = (unsigned)(logical >> 2);        // which tap of the pair this lane's piece belongs to
  const int brow = n0 + lr;
  const unsigned b_voff = (unsigned)((long long)brow * p.Cin * 2) + (logical & 3) * 16;
  const unsigned w_tap_bytes = (unsigned)((long long)p.wrows * p.Cin * 2);
  const unsigned b_row_step = (unsigned)(64 * p.Cin * 2);
  auto issue_b = [&](int buf, int chunk, int ks) {       // taps 2 ks and 2 ks + 1 (tap 27 does not exist: zeros)
    const unsigned off0 = (unsigned)(2 * ks) * w_tap_bytes + (unsigned)(chunk * 64);
    const bool two = 2 * ks + 1 < 27;
    unsigned sel_off = b_sel ? (two ? off0 + w_tap_bytes : kOOB) : off0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned step = b_row_step * i;
      asm volatile("" : "+s"(step));                      // opaque scalar: the address is formed here (not hoisted 56-fold out of the chunk loop)
      const bool ok = brow + 64 * i < p.wrows && sel_off != kOOB;
      lds_dma16(wr, Bb + buf * B_BYTES + (wave_u * 8 + 64 * i) * 128, ok ? b_voff + sel_off + step : kOOB);
    }
  };

  // XP: the two 64-byte runs of a weight-tile row come from two arbitrary (tap, chunk) entries
  auto issue_b2 = [&](int buf, int chunk_a, int tap_a, int chunk_b, int tap_b) {
    const unsigned off_a = (unsigned)tap_a * w_tap_bytes + (unsigned)(chunk_a * 64);
    const unsigned off_b = (unsigned)tap_b * w_tap_bytes + (unsigned)(chunk_b * 64);
    const unsigned sel_off = b_sel ? off_b : off_a;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned step = b_row_step * i;
      asm volatile("" : "+s"(step));
      const bool ok = brow + 64 * i < p.wrows;
      lds_dma16(wr, Bb + buf * B_BYTES + (wave_u * 8 + 64 * i) * 128, ok ? b_voff + sel_off + step : kOOB);
    }
  };

  // ---- fragment addresses
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 31, fk = lane >> 5;
  int a_base[TM], b_addr[TN][4];
  {
    int xl, yl, zl;
    block_voxel(fr, xl, yl, zl);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int blk = wm * TM + i;                       // 8 blocks: x pair (blk >> 2), y pair (blk & 3)
      a_base[i] = ((2 * (blk >> 2) + xl) * PX + (2 * (blk & 3) + yl) * PY + zl) * 16 + fk * (RP * 16);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = (wn * TN + j) * 32 + fr;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) b_addr[j][s4] = 2 * A_BYTES + n * 128 + (((s4 * 2 + fk) ^ ((n >> 1) & 7)) << 4);
    }
  }
  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunk = p.Cin / 32;
  // prologue: halo of chunk 0 and the first weight tile
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) issue_halo_piece(0, 0, j);
  issue_b(0, 0, 0);
  __syncthreads();

  // one K-step = taps (2 ks, 2 ks + 1) x 32 channels: sub-steps (tap, channel half) = (0,0) (0,1) (1,0) (1,1).  Rotated pipeline as in
  // conv_igemm_big_kernel: the fragments of a sub-step are loaded under the MFMAs of the one before; the barrier that hands over the next
  // weight tile sits BEFORE the MFMAs of a K-step's last sub-step, whose fragments are already in registers, and the first fragments of
  // the NEXT K-step are fetched right behind it.
  f4 af[2][TM], bfv[2][TN];
  auto load = [&](int abuf, int bbuf, int ks, int s4, f4 (&a)[TM], f4 (&b)[TN]) {
    const int tap = 2 * ks + (s4 >> 1);
    const int dxyz = (tap / 9) * PX + ((tap / 3) % 3) * PY + (tap % 3);
    const int aoff = abuf * A_BYTES + (s4 & 1) * (2 * RP * 16) + dxyz * 16;
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f4 *>(lds + b_addr[j][s4] + bbuf * B_BYTES);
#pragma unroll
    for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f4 *>(lds + a_base[i] + aoff);
  };
  auto kstep = [&](int abuf, int ks, int chunk, bool more_chunks) {
    const int bbuf = ks & 1;
    const int subs = (2 * ks + 1 < 27) ? 4 : 2;
    const bool last = ks + 1 == KSTEPS;
    // the next weight tile (and, early in a chunk, the next chunk's halo) fly under this K-step's MFMAs
    auto issue_next_b = [&]() {
      if (!last) issue_b(bbuf ^ 1, chunk, ks + 1);
      else if (more_chunks) issue_b(bbuf ^ 1, chunk + 1, 0);
    };
    const bool late = (V & 2) && wave_u >= 4;
    if (!(V & 2)) issue_next_b();
    else if (!late) issue_next_b();
    if (more_chunks && ks < A_PER_WAVE) issue_halo_piece(abuf ^ 1, chunk + 1, ks);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      if (s4 >= subs) break;
      if ((V & 2) && s4 == 1 && late) issue_next_b();
      if (s4 + 1 < subs) {
        load(abuf, bbuf, ks, s4 + 1, af[(s4 + 1) & 1], bfv[(s4 + 1) & 1]);
      } else {
        __syncthreads();
        if (!last) load(abuf, bbuf ^ 1, ks + 1, 0, af[0], bfv[0]);
        else if (more_chunks) load(abuf ^ 1, bbuf ^ 1, 0, 0, af[0], bfv[0]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[s4 & 1][i], bfv[s4 & 1][j]);
#pragma unroll
      for (int q = 0; q < TM + TN; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
    }
  };
  load(0, 0, 0, 0, af[0], bfv[0]);
  if ((V & 1) && wave_u >= 4) __builtin_amdgcn_s_setprio(1);
  if constexpr (XP) {
    // entry e (0..107) of a four-chunk group = (chunk e / 27, tap e % 27); K-step k = entries 2k, 2k + 1; sub-steps (entry, channel half)
    auto load_xp = [&](int k, int s4, f4 (&a)[TM], f4 (&b)[TN]) {
      const int e = 2 * k + (s4 >> 1), cj = e / 27, tap = e % 27;
      const int dxyz = (tap / 9) * PX + ((tap / 3) % 3) * PY + (tap % 3);
      const int aoff = (cj & 1) * A_BYTES + (s4 & 1) * (2 * RP * 16) + dxyz * 16;
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f4 *>(lds + b_addr[j][s4] + (k & 1) * B_BYTES);
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f4 *>(lds + a_base[i] + aoff);
    };
    auto kstep_xp = [&](auto kc, int chunk, bool more) {
      constexpr int k = decltype(kc)::value;
      constexpr int bbuf = k & 1;
      constexpr bool last = k == 53;
      if constexpr (!last) {
        constexpr int e0 = 2 * k + 2, e1 = e0 + 1;
        issue_b2(bbuf ^ 1, chunk + e0 / 27, e0 % 27, chunk + e1 / 27, e1 % 27);
      } else if (more) {
        issue_b2(bbuf ^ 1, chunk + 4, 0, chunk + 4, 1);
      }
      // halo of chunk j + 1 goes into the buffer chunk j - 1 used: free once the K-step holding chunk j - 1's last entry has passed its barrier
      // (chunk 0 / 1 / 2 / 3 of the group end in K-steps 13 / 26 / 40 / 53)
      if constexpr (k < A_PER_WAVE) issue_halo_piece(1, chunk + 1, k);
      else if constexpr (k >= 14 && k < 14 + A_PER_WAVE) issue_halo_piece(0, chunk + 2, k - 14);
      else if constexpr (k >= 27 && k < 27 + A_PER_WAVE) issue_halo_piece(1, chunk + 3, k - 27);
      else if constexpr (k >= 41 && k < 41 + A_PER_WAVE) { if (more) issue_halo_piece(0, chunk + 4, k - 41); }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        if (s4 < 3) {
          load_xp(k, s4 + 1, af[(s4 + 1) & 1], bfv[(s4 + 1) & 1]);
        } else {
          __syncthreads();
          if constexpr (!last) load_xp(k + 1, 0, af[0], bfv[0]);
          else if (more) load_xp(0, 0, af[0], bfv[0]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[s4 & 1][i], bfv[s4 & 1][j]);
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
      }
    };
#pragma unroll 1
    for (int chunk = 0; chunk < nchunk; chunk += 4) {
      const bool more = chunk + 4 < nchunk;
      static_for<0, 54>([&](auto kc) { kstep_xp(kc, chunk, more); });
    }
  } else {
  // chunks in pairs so that the halo buffer is a compile-time constant inside the unrolled K-steps (immediate ds offsets); 14 K-steps
  // per chunk flip the weight buffer an even number of times, so every chunk starts on weight buffer 0
#pragma unroll 1
  for (int chunk = 0; chunk < nchunk; chunk += 2) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) kstep(0, ks, chunk, chunk + 1 < nchunk);
    if (chunk + 1 < nchunk) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) kstep(1, ks, chunk + 1, chunk + 2 < nchunk);
    }
  }
  }

  // ---- epilogue: scale / bias / ReLU / optional ReLU mask / optional BatchNorm statistics; staged through LDS, 16-byte stores
  const bool has_bias = (p.flags & NRPN_CONV_BIAS) && p.bias;
  const bool relu = p.flags & NRPN_CONV_RELU;
  int elane = lane;
  asm volatile("" : "+v"(elane));
  const int efr = elane & 31;
  constexpr int PITCH = 144;
  char *stage = lds + wave * (64 * PITCH);
  const T *maskp = reinterpret_cast<const T *>(p.mask);
  T *yp = reinterpret_cast<T *>(p.y);
  float ssum[TN] = {0.f, 0.f}, qsum[TN] = {0.f, 0.f};
  const bool full_block = x0 + TX <= p.X && y0 + TY <= p.Y && z0 + TZ <= p.Z;
  if constexpr (OF32) {
    // fp32 rows: the same staging with 4-byte elements (64 rows x 64 columns per pass = 256-byte rows, pitch 272), 16-byte stores
    constexpr int PITCH32 = 272;
    char *stage32 = lds + wave * (64 * PITCH32);
    float *yf = reinterpret_cast<float *>(p.y);
    __syncthreads();                                   // the staging regions overlap other waves' K-loop buffers
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + (wn * TN + j) * 32 + efr;
        const float bv = (has_bias && col < p.Cout) ? p.bias[col] : 0.f;
        const float sv = (p.scale && col < p.Cout) ? p.scale[col] : 1.f;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = half * 2 + ii;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = frag_row(r, elane);
            float o = acc[i][j][r] * sv + bv;
            if (relu) o = fmaxf(o, 0.f);
            *reinterpret_cast<float *>(stage32 + (ii * 32 + rr) * PITCH32 + (j * 32 + efr) * 4) = o;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int pc = elane + 64 * q;                  // 64 rows x 16 pieces of 4 floats
        const int row = pc >> 4, seg = pc & 15;
        const int blk = wm * TM + half * 2 + (row >> 5);
        int xl, yl, zl;
        block_voxel(row & 31, xl, yl, zl);
        const int gx = x0 + 2 * (blk >> 2) + xl, gy = y0 + 2 * (blk & 3) + yl, gz = z0 + zl;
        const int col = n0 + wn * 64 + seg * 4;
        if (gx < p.X && gy < p.Y && gz < p.Z && col < p.Cout) {
          const long long v = vbase + ((long long)gx * p.Y + gy) * p.Z + gz;
          *reinterpret_cast<f4 *>(yf + v * p.Cout + col) = *reinterpret_cast<const f4 *>(stage32 + row * PITCH32 + seg * 16);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + efr;
      const float bv = (has_bias && col < p.Cout) ? p.bias[col] : 0.f;
      const float sv = (p.scale && col < p.Cout) ? p.scale[col] : 1.f;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = half * 2 + ii, blk = wm * TM + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = frag_row(r, elane);
          float o = acc[i][j][r] * sv + bv;
          if (relu) o = fmaxf(o, 0.f);
          const bf16s ob = f32_to_bf16_bits(o);
          *reinterpret_cast<bf16s *>(stage + (ii * 32 + rr) * PITCH + (j * 32 + efr) * 2) = ob;
          if (p.stats) {
            bool in = true;
            if (!full_block) {        // ragged block at the grid's edge: only voxels inside the grid count (wave-uniform branch)
              int xl, yl, zl;
              block_voxel(rr, xl, yl, zl);
              in = x0 + 2 * (blk >> 2) + xl < p.X && y0 + 2 * (blk & 3) + yl < p.Y && z0 + zl < p.Z;
            }
            const float of = in ? bf16_bits_to_f32(ob) : 0.f;
            ssum[j] += of;
            qsum[j] += of * of;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int pc = elane + 64 * q;                  // 64 rows x 8 pieces
      const int row = pc >> 3, seg = pc & 7;
      const int blk = wm * TM + half * 2 + (row >> 5);
      int xl, yl, zl;
      block_voxel(row & 31, xl, yl, zl);
      const int gx = x0 + 2 * (blk >> 2) + xl, gy = y0 + 2 * (blk & 3) + yl, gz = z0 + zl;
      const int col = n0 + wn * 64 + seg * 8;
      if (gx < p.X && gy < p.Y && gz < p.Z && col < p.Cout) {
        const long long v = vbase + ((long long)gx * p.Y + gy) * p.Z + gz;
        f4 val = *reinterpret_cast<const f4 *>(stage + row * PITCH + seg * 16);
        if (maskp) {
          typedef __attribute__((ext_vector_type(8))) unsigned short u8v;
          const u8v mk = __builtin_bit_cast(u8v, *reinterpret_cast<const f4 *>(maskp + v * p.Cout + col));
          u8v ov = __builtin_bit_cast(u8v, val);
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = (bf16_bits_to_f32(mk[e]) > 0.f) ? ov[e] : (unsigned short)0;
          val = __builtin_bit_cast(f4, ov);
        }
        *reinterpret_cast<f4 *>(yp + v * p.Cout + col) = val;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (p.stats) {
    const long long prow = (long long)(tile / ntiles) * 2 + wm;
#pragma unroll
    for (int j = 0; j < TN; ++j) store_col_stats(p.stats, prow, p.Cout, n0 + (wn * TN + j) * 32 + efr, ssum[j], qsum[j], elane);
  }
}


// variant: bits 0-1 = V (tools-only A/B variants of the classic K order), bit 2 = XP (cross-chunk tap pairing; the caller guarantees
// Cin % 128 == 0), bit 3 = OF32 (fp32 output rows)
int nrpn_launch_conv_halo(const ConvArgs &a, unsigned workgroups, hipStream_t st, int variant) {
#define NRPN_HALO(...)                                                                                                \
  do {                                                                                                                \
    NRPN_LDS((conv_halo_kernel<__VA_ARGS__>), hk::LDS_BYTES);                                                         \
    hipLaunchKernelGGL((conv_halo_kernel<__VA_ARGS__>), dim3(workgroups), dim3(512), hk::LDS_BYTES, st, a);           \
  } while (0)
  const bool xp = (variant & 4) != 0, of32 = (variant & 8) != 0;
  if (of32) {
    if (xp) NRPN_HALO(0, true, true); else NRPN_HALO(0, false, true);
  } else if (xp) {
    NRPN_HALO(0, true, false);
  } else {
    switch (variant & 3) {
      case 1: NRPN_HALO(1); break;
      case 2: NRPN_HALO(2); break;
      case 3: NRPN_HALO(3); break;
      default: NRPN_HALO(0); break;
    }
  }
#undef NRPN_HALO
  NRPN_LAUNCH_CHECK("conv_halo");
  return NRPN_OK;
}

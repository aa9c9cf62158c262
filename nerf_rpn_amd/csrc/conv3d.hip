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
#include "conv_common.cuh"

// Halo kernel, taps paired across channel-chunk boundaries (conv_halo.hip XP): 0 = never, 1 = whenever Cin % 128 == 0, 2 = from Cin 256 up.
// Measured (profiles/r05_halo_pairing.json, one MI355X, alternating rounds, bit-identical outputs): 256->256@40^3 176.1 vs 176.5 us (post-ReLU
// operands) / 181.5 vs 182.0 us (random) -- the half-empty 14th K-step was not what the kernel waits for; 128->256 (one four-chunk group) 96.9 vs
// 96.8 / 100.1 vs 99.9 us; the bf16x3 form of 256->256 (Cin 768: six groups, fp32 rows) 496 vs 525 us (-5.5 %).
#ifndef NRPN_HALO_PAIRING_DEFAULT
#define NRPN_HALO_PAIRING_DEFAULT 2
#endif

// ROWS (MODE 0): row-list form -- tile row v is voxel p.rows[2 v] of the ragged space with tap word p.rows[2 v + 1] (csrc/cone.hip); the
// output (and the optional ReLU mask) row is that voxel.  The dense instantiations carry none of it.
template <typename T, int BN, int MODE, bool OUTF32, int KB, bool GLDS, int BM = 128, bool ROWS = false>
__global__ void __launch_bounds__(256, BM == 128 ? 2 : 1) conv_igemm_kernel(const ConvArgs p) {
  constexpr int WAVES_N = (BN == 128) ? 2 : 1;
  constexpr int TM = BM / (32 * (4 / WAVES_N));   // 32x32 tiles per wave along M: 1, 2 or (BM 256) 4
  constexpr int TN = 2;                            // ... along N
  constexpr int KE = KB / (int)sizeof(T);    // K elements per step
  constexpr int PPR = KB / 16;               // 16-byte pieces per tile row (4 or 8)
  constexpr int RSTEP = 256 / PPR;           // row distance between a thread's pieces
  constexpr int A_RPT = BM / RSTEP;          // A rows per thread (2 or 4)
  constexpr int B_RPT = BN / RSTEP;          // B rows per thread (1, 2 or 4)
  constexpr int SWZ_SH = (KB == 64) ? 2 : 1; // rows per 256-byte LDS bank row = 256 / KB
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB;
  extern __shared__ __attribute__((aligned(16))) char lds[];      // 2 * (A_BYTES + B_BYTES), sized by the launcher

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const unsigned ntiles = (p.Cout + BN - 1) / BN;
  const unsigned tiles = (unsigned)((p.M + BM - 1) / BM) * ntiles;
  // whole tiles, every tile on `ksplit` K slices, or (tail_ks > 1) only the M tiles from tail_tile0 on -- the short last round of a launch
  // that needs a little more than a whole number of rounds of the chip -- on tail_ks slices (see conv_igemm_big_kernel)
  unsigned zsplit, tile;
  int my_ks = p.ksplit > 1 ? p.ksplit : 1;
  long long part_row0 = 0;
  if (p.tail_ks > 1) {
    const unsigned full = (unsigned)p.tail_tile0 * ntiles;
    if (blockIdx.x < full) {
      zsplit = 0; my_ks = 1;
      tile = xcd_remap(blockIdx.x, full);
    } else {
      const unsigned t = blockIdx.x - full, ntail = tiles - full;
      zsplit = t / ntail; my_ks = p.tail_ks;
      tile = full + t % ntail;
      part_row0 = (long long)p.tail_tile0 * BM;
    }
  } else {
    zsplit = blockIdx.x / tiles;                                   // split-K slice (0 when ksplit == 1)
    tile = xcd_remap(blockIdx.x - zsplit * tiles, tiles);
  }
  const long long m0 = (long long)(tile / ntiles) * BM;
  const int n0 = (int)(tile % ntiles) * BN;
  const int lr = tid / PPR, ls = tid % PPR;

  // ---- loader state per A row: element offset of the voxel, per-scene coordinates, and a 27-bit mask of in-bounds taps
  long long a_off[A_RPT];
  int a_x[A_RPT], a_y[A_RPT], a_z[A_RPT];
  bool a_ok[A_RPT];
  unsigned a_mask[A_RPT];
  int a_yz[A_RPT], a_zs[A_RPT];   // MODE 0: byte strides of one x / one y step in this row's own grid (ragged lists mix grids)
  long long a_nbase[A_RPT];  // MODE 1: element offset of batch n
#pragma unroll
  for (int i = 0; i < A_RPT; ++i) {
    const long long v = m0 + lr + RSTEP * i;
    a_ok[i] = v < p.M;
    const long long vv = a_ok[i] ? v : 0;
    int ox, oy, oz, gX = p.X, gY = p.Y, gZ = p.Z;
    long long n = 0;
    unsigned row_word = 0;
    long long row_vox = vv;
    if (MODE == 0 && ROWS) {
      if (a_ok[i]) { row_vox = p.rows[2 * vv]; row_word = p.rows[2 * vv + 1]; }
      const int seg = (int)(row_word >> 27);
#pragma unroll
      for (int q = 0; q < kMaxSeg; ++q)
        if (q == seg && q < p.segs.n) { gY = p.segs.Y[q]; gZ = p.segs.Z[q]; }
      ox = oy = oz = 0;
    } else if (MODE == 0) {
      if (p.segs.n == 0) {       // classic layout: multiply-shift divisions (M < 2^31 is checked by the launcher)
        const unsigned u = (unsigned)vv, sc = fastdiv(u, p.dvs), local = u - sc * (unsigned)(p.X * p.Y * p.Z), t1 = fastdiv(local, p.dvz);
        oz = (int)(local - t1 * (unsigned)p.Z);
        ox = (int)fastdiv(t1, p.dvy);
        oy = (int)(t1 - (unsigned)ox * (unsigned)p.Y);
      } else {
        locate_voxel(p.segs, vv, p.X, p.Y, p.Z, ox, oy, oz, gX, gY, gZ);
      }
    } else {
      oz = (int)(vv % p.OZ);
      const long long t1 = vv / p.OZ;
      oy = (int)(t1 % p.OY);
      const long long t2 = t1 / p.OY;
      ox = (int)(t2 % p.OX);
      n = t2 / p.OX;
    }
    a_x[i] = ox; a_y[i] = oy; a_z[i] = oz;
    a_yz[i] = gY * gZ * p.Cin * (int)sizeof(T);
    a_zs[i] = gZ * p.Cin * (int)sizeof(T);
    a_off[i] = (MODE == 0) ? row_vox * p.Cin : 0;
    a_nbase[i] = (MODE == 0) ? 0 : n * (long long)p.X * p.Y * p.Z * 4;
    unsigned m = 0;
    if (MODE == 0 && ROWS) {
      m = p.taps == 27 ? (row_word & 0x7FFFFFFu) : (a_ok[i] ? 1u : 0u);
    } else if (MODE == 0 && a_ok[i]) {
      m = p.taps == 27 ? tap_mask27(ox, oy, oz, gX, gY, gZ) : 1u;
    }
    a_mask[i] = m;
  }
  const T *wbase = reinterpret_cast<const T *>(p.w);
  const T *xbase = reinterpret_cast<const T *>(p.x);

  const int cpt = (MODE == 0) ? p.Cin / KE : 1;                                   // K-steps per tap
  const int nk = (MODE == 0) ? p.taps * cpt : (p.taps * 4 + KE - 1) / KE;        // total K-steps
  const int kpad = nk * KE;                                                        // MODE 1 weight row length

  // split-K: this workgroup runs K-steps [ks_begin, ks_end)
  int ks_begin = 0, ks_end = nk;
  if (my_ks > 1) {   // the launcher trims the slice count so that no slice is empty (every partial is fully written)
    const int per = (nk + my_ks - 1) / my_ks;
    ks_begin = zsplit * per;
    ks_end = min(nk, ks_begin + per);
  }

  // MODE 0: byte offsets for the raw-buffer loads (loop invariant); kOOB marks rows that must read zeros
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
  unsigned a_voff[A_RPT], b_voff[B_RPT];
#pragma unroll
  for (int i = 0; i < A_RPT; ++i) {
    // LDS-DMA writes lane l of a wave to (base + 16 l): the physical 16-byte slot is fixed by the lane, so the XOR swizzle
    // is applied to the SOURCE slot instead (linear destination + inverse-swizzled source + swizzled read)
    const int r = lr + RSTEP * i;
    const int src_slot = GLDS ? (ls ^ ((r >> SWZ_SH) & (PPR - 1))) : ls;
    a_voff[i] = (unsigned)(a_off[i] * (long long)sizeof(T)) + src_slot * 16;
  }
#pragma unroll
  for (int i = 0; i < B_RPT; ++i) {
    const int row = n0 + lr + RSTEP * i;
    const int rb_ = lr + RSTEP * i;
    const int src_slot = GLDS ? (ls ^ ((rb_ >> SWZ_SH) & (PPR - 1))) : ls;
    b_voff[i] = row < p.wrows ? (unsigned)((long long)row * p.Cin * (long long)sizeof(T)) + src_slot * 16 : kOOB;
  }
  f4 ra0[A_RPT], rb0[B_RPT];
  // incremental (tap, chunk) walk of the K loop for MODE 0 -- no divisions in the steady state
  int l_tap = ks_begin / cpt, l_chunk = ks_begin - l_tap * cpt;
  int l_dx = 0, l_dy = 0, l_dz = 0;
  if (p.taps == 27) { l_dx = l_tap / 9 - 1; l_dy = (l_tap / 3) % 3 - 1; l_dz = l_tap % 3 - 1; }
  const long long w_tap_stride = (long long)p.wrows * p.Cin;

  auto load_step = [&](int ks, f4 (&ra)[A_RPT], f4 (&rb)[B_RPT]) {
    if (MODE == 0) {
      const int c0 = l_chunk * KE;
      const int zshift = (l_dz * p.Cin + c0) * (int)sizeof(T);
#pragma unroll
      for (int i = 0; i < A_RPT; ++i)
        ra[i] = bufld16(xr, ((a_mask[i] >> l_tap) & 1u) ? a_voff[i] + (unsigned)(l_dx * a_yz[i] + l_dy * a_zs[i] + zshift) : kOOB);
      const unsigned wshift = (unsigned)(((long long)l_tap * w_tap_stride + c0) * (long long)sizeof(T));
#pragma unroll
      for (int i = 0; i < B_RPT; ++i) rb[i] = bufld16(wr, b_voff[i] + wshift);
      if (++l_chunk == cpt) {
        l_chunk = 0;
        ++l_tap;
        if (++l_dz > 1) { l_dz = -1; if (++l_dy > 1) { l_dy = -1; ++l_dx; } }
      }
    } else {
      // stem: slot ls of K-step ks covers taps [t0, t0 + TPS), 4 input channels each
      constexpr int TPS = 16 / (4 * (int)sizeof(T));  // taps per 16-byte slot: 1 (fp32) or 2 (bf16)
#pragma unroll
      for (int i = 0; i < A_RPT; ++i) {
        f4 v = zero4();
#pragma unroll
        for (int q = 0; q < TPS; ++q) {
          const int tap = (ks * PPR + ls) * TPS + q;
          const int dx = tap / 49, dy = (tap / 7) % 7, dz = tap % 7;
          const int ix = a_x[i] * p.stride - 3 + dx, iy = a_y[i] * p.stride - 3 + dy, iz = a_z[i] * p.stride - 3 + dz;
          const bool in = a_ok[i] && tap < p.taps && (unsigned)ix < (unsigned)p.X && (unsigned)iy < (unsigned)p.Y &&
                          (unsigned)iz < (unsigned)p.Z;
          const unsigned off = in ? (unsigned)((a_nbase[i] + (((long long)ix * p.Y + iy) * p.Z + iz) * 4) * (long long)sizeof(T)) : kOOB;
          if (sizeof(T) == 4) v = bufld16(xr, off);
          else {
            typedef __attribute__((ext_vector_type(2))) unsigned int u2v;
            const u2v h = __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0);
            v[2 * q] = __uint_as_float(h[0]); v[2 * q + 1] = __uint_as_float(h[1]);
          }
        }
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_RPT; ++i) {
        const int row = n0 + lr + RSTEP * i;
        rb[i] = (row < p.wrows) ? ldg16(reinterpret_cast<const char *>(wbase + (long long)row * kpad + ks * KE) + ls * 16) : zero4();
      }
    }
  };

  // LDS-DMA form of load_step (MODE 0): buffer_load_dwordx4 ... lds straight into LDS buffer `buf`, no staging registers and
  // no ds_write.  A wave-instruction covers 64 / PPR consecutive tile rows = 1 KiB of LDS at a wave-uniform base.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_glds = [&](int buf) {
    char *A = lds + buf * (A_BYTES + B_BYTES);
    char *B = A + A_BYTES;
    const int c0 = l_chunk * KE;
    const int zshift = (l_dz * p.Cin + c0) * (int)sizeof(T);
    const unsigned wshift = (unsigned)(((long long)l_tap * w_tap_stride + c0) * (long long)sizeof(T));
#pragma unroll
    for (int i = 0; i < A_RPT; ++i)
      lds_dma16(xr, A + (wave_u * (64 / PPR) + RSTEP * i) * KB,
                ((a_mask[i] >> l_tap) & 1u) ? a_voff[i] + (unsigned)(l_dx * a_yz[i] + l_dy * a_zs[i] + zshift) : kOOB);
#pragma unroll
    for (int i = 0; i < B_RPT; ++i)
      lds_dma16(wr, B + (wave_u * (64 / PPR) + RSTEP * i) * KB, b_voff[i] + wshift);
    if (++l_chunk == cpt) {
      l_chunk = 0;
      ++l_tap;
      if (++l_dz > 1) { l_dz = -1; if (++l_dy > 1) { l_dy = -1; ++l_dx; } }
    }
  };

  auto store_step = [&](int buf, const f4 (&ra)[A_RPT], const f4 (&rb)[B_RPT]) {
    char *A = lds + buf * (A_BYTES + B_BYTES);
    char *B = A + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_RPT; ++i) {
      const int r = lr + RSTEP * i;
      *reinterpret_cast<f4 *>(A + r * KB + ((ls ^ ((r >> SWZ_SH) & (PPR - 1))) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_RPT; ++i) {
      const int r = lr + RSTEP * i;
      *reinterpret_cast<f4 *>(B + r * KB + ((ls ^ ((r >> SWZ_SH) & (PPR - 1))) << 4)) = rb[i];
    }
  };

  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  auto compute_step = [&](int buf) {
    const char *A = lds + buf * (A_BYTES + B_BYTES);
    const char *B = A + A_BYTES;
#pragma unroll
    for (int s = 0; s < KB / 32; ++s) {
      f4 af[TM], bfv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int r = (wm * TM + i) * 32 + fr;
        af[i] = *reinterpret_cast<const f4 *>(A + r * KB + (((s * 2 + fk) ^ ((r >> SWZ_SH) & (PPR - 1))) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int r = (wn * TN + j) * 32 + fr;
        bfv[j] = *reinterpret_cast<const f4 *>(B + r * KB + (((s * 2 + fk) ^ ((r >> SWZ_SH) & (PPR - 1))) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[i], bfv[j]);
    }
  };

  // Software pipeline: while step k is multiplied out of LDS buffer k&1, the loads of step k+1 are in flight into the
  // staging registers and are written to the other buffer after the MFMAs.  The loop is unrolled by two so both LDS
  // buffers are compile-time constants (immediate ds offsets, no address arithmetic in the loop).
  int ks = ks_begin;
  if (GLDS && MODE == 0) {
    // 2-buffer LDS-DMA pipeline: the DMA of step k+1 flies while step k is multiplied; __syncthreads() carries the vmcnt(0)
    issue_glds(0);
    __syncthreads();
    // pairs of K-steps with ONE exit at the top of the loop (an exit in the middle made the compiler keep the accumulators in two register
    // sets and copy all 64 of them once per pair: 32 v_mov_b64 + s_nop bubbles behind the MFMAs, tools/isa_audit.py); the odd last step follows
    for (; ks + 1 < ks_end; ks += 2) {
      issue_glds(1);
      compute_step(0);
      __syncthreads();
      if (ks + 2 < ks_end) issue_glds(0);
      compute_step(1);
      __syncthreads();
    }
    if (ks < ks_end) {
      compute_step(0);
      __syncthreads();
    }
  } else {
    load_step(ks_begin, ra0, rb0);
    store_step(0, ra0, rb0);
    __syncthreads();
    for (; ks + 1 < ks_end; ks += 2) {        // pairs of K-steps, one loop exit (see the LDS-DMA pipeline above)
      load_step(ks + 1, ra0, rb0);
      compute_step(0);
      store_step(1, ra0, rb0);
      __syncthreads();
      if (ks + 2 < ks_end) load_step(ks + 2, ra0, rb0);
      compute_step(1);
      if (ks + 2 < ks_end) store_step(0, ra0, rb0);
      __syncthreads();
    }
    if (ks < ks_end) {
      compute_step(0);
      __syncthreads();
    }
  }

  if (my_ks > 1) {   // fp32 partial of this K slice, plain stores (deterministic); the slices are summed in order, then bias /
                     // ReLU / cast, by splitk_epilogue_kernel / rows_epilogue_kernel
    float *wsz = p.ws + (long long)zsplit * (p.M - part_row0) * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + fr;
      if (col >= p.Cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long v = m0 + (wm * TM + i) * 32 + frag_row(r, lane);
          if (v < p.M) wsz[(v - part_row0) * p.Cout + col] = acc[i][j][r];
        }
    }
    return;
  }

  // ---- epilogue: bias / ReLU, store channels-last
  const bool has_bias = (p.flags & NRPN_CONV_BIAS) && p.bias;
  const bool relu = p.flags & NRPN_CONV_RELU;
  constexpr int PITCH = 144;
  if (sizeof(T) == 2 && !OUTF32 && 2 * (A_BYTES + B_BYTES) >= 4 * TM * 32 * PITCH && (p.Cout & 7) == 0) {
    // bf16 output: stage the wave's (TM*32) x 64 block through LDS (free after the loop's last barrier) and store / mask it as
    // whole 16-byte pieces instead of one 2-byte element per lane per row (see conv_igemm_big_kernel)
    char *stage = lds + wave * (TM * 32 * PITCH);
    const T *maskp = reinterpret_cast<const T *>(p.mask);
    T *yp = reinterpret_cast<T *>(p.y);
    const bool full_tile = m0 + BM <= p.M;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + fr;
      const float bv = (has_bias && col < p.Cout) ? p.bias[col] : 0.f;
      const float sv = (p.scale && col < p.Cout) ? p.scale[col] : 1.f;
      float ssum = 0.f, qsum = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float o = acc[i][j][r] * sv + bv;
          if (relu) o = fmaxf(o, 0.f);
          const bf16s ob = f32_to_bf16_bits(o);
          *reinterpret_cast<bf16s *>(stage + (i * 32 + frag_row(r, lane)) * PITCH + (j * 32 + fr) * 2) = ob;
          if (p.stats && (full_tile || m0 + (wm * TM + i) * 32 + frag_row(r, lane) < p.M)) {
            const float of = bf16_bits_to_f32(ob);
            ssum += of;
            qsum += of * of;
          }
        }
      if (p.stats) store_col_stats(p.stats, (m0 / BM) * (BM / (TM * 32)) + wm, p.Cout, col, ssum, qsum, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < TM * 4; ++q) {
      const int pc = lane + 64 * q;                     // TM*32 rows x 8 pieces of 16 bytes
      const int row = pc >> 3, seg = pc & 7;
      const long long vt = m0 + wm * (TM * 32) + row;
      const int col = n0 + wn * 64 + seg * 8;
      if (vt < p.M && col < p.Cout) {
        const long long v = ROWS ? (long long)p.rows[2 * vt] : vt;
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
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + fr;
    if (col >= p.Cout) continue;
    const float bv = has_bias ? p.bias[col] : 0.f;
    const float sv = p.scale ? p.scale[col] : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long vt = m0 + (wm * TM + i) * 32 + frag_row(r, lane);
        if (vt < p.M) {
          const long long v = ROWS ? (long long)p.rows[2 * vt] : vt;
          float o = acc[i][j][r] * sv + bv;
          if (relu) o = fmaxf(o, 0.f);
          if (p.mask && !(elem<T>::ld(reinterpret_cast<const T *>(p.mask) + v * p.Cout + col) > 0.f)) o = 0.f;
          if (OUTF32) reinterpret_cast<float *>(p.y)[v * p.Cout + col] = o;
          else elem<T>::st(reinterpret_cast<T *>(p.y) + v * p.Cout + col, o);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant for the heavy bf16 shapes (k1 / k3, Cin*2 % 128 == 0, enough 256x128 tiles to fill the chip).
// 512 threads = 8 waves, two per SIMD: waves 0-3 are CONSUMERS (each owns a 128x64 block of the 256x128 tile: 8 accumulators,
// 6 ds_read_b128 per 8 MFMAs) and never touch global memory; waves 4-7 are PRODUCERS that only issue the LDS-DMA pieces of the
// next K-step (12 per wave per step) -- the ~100-cycle issue cost of a `buffer_load ... lds` no longer sits in the MFMA wave's
// instruction stream, and the bigger per-wave block cuts LDS reads and DMA pieces per MFMA by 25 %.  Three LDS buffers (144 KB), one barrier per
// K-step, DMA issued two steps ahead.
// ---------------------------------------------------------------------------------------------------------------------
template <bool OUTF32>
__global__ void __launch_bounds__(512, 1) conv_igemm_ws_kernel(const ConvArgs p) {
  typedef bf16s T;
  constexpr int BM = 256, BN = 128, KB = 128, KE = 64, PPR = 8, RSTEP = 32, TM = 4, TN = 2;
  constexpr int A_RPT = BM / RSTEP, B_RPT = BN / RSTEP;    // 8 + 4 DMA pieces per producer lane per K-step
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const unsigned ntiles = (p.Cout + BN - 1) / BN;
  const unsigned tiles = (unsigned)((p.M + BM - 1) / BM) * ntiles;
  const unsigned tile = xcd_remap(blockIdx.x, tiles);
  const long long m0 = (long long)(tile / ntiles) * BM;
  const int n0 = (int)(tile % ntiles) * BN;
  const int cpt = p.Cin / KE;
  const int nk = p.taps * cpt;

  if (producer) {
    const int ptid = tid - 256, lr = ptid / PPR, ls = ptid % PPR;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave - 4);
    const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
    unsigned a_voff[A_RPT], a_mask[A_RPT], b_voff[B_RPT];
#pragma unroll
    for (int i = 0; i < A_RPT; ++i) {
      const int r = lr + RSTEP * i;
      const long long v = m0 + r;
      const bool ok = v < p.M;
      const long long vv = ok ? v : 0;
      const int oz = (int)(vv % p.OZ);
      const long long t1 = vv / p.OZ;
      const int oy = (int)(t1 % p.OY);
      const int ox = (int)((t1 / p.OY) % p.OX);
      unsigned m = 0;
      if (ok) {
        if (p.taps == 27) {
#pragma unroll
          for (int t = 0; t < 27; ++t) {
            const int dx = t / 9 - 1, dy = (t / 3) % 3 - 1, dz = t % 3 - 1;
            const bool in = (unsigned)(ox + dx) < (unsigned)p.X && (unsigned)(oy + dy) < (unsigned)p.Y && (unsigned)(oz + dz) < (unsigned)p.Z;
            m |= in ? (1u << t) : 0u;
          }
        } else {
          m = 1u;
        }
      }
      a_mask[i] = m;
      a_voff[i] = (unsigned)(vv * p.Cin * 2) + ((ls ^ ((r >> 1) & (PPR - 1))) << 4);    // inverse-swizzled source slot
    }
#pragma unroll
    for (int i = 0; i < B_RPT; ++i) {
      const int r = lr + RSTEP * i, row = n0 + r;
      b_voff[i] = row < p.wrows ? (unsigned)((long long)row * p.Cin * 2) + ((ls ^ ((r >> 1) & (PPR - 1))) << 4) : kOOB;
    }
    int l_tap = 0, l_chunk = 0, l_dx = 0, l_dy = 0, l_dz = 0;
    if (p.taps == 27) { l_dx = -1; l_dy = -1; l_dz = -1; }
    const long long w_tap_stride = (long long)p.wrows * p.Cin;
    auto issue = [&](int buf) {
      char *A = lds + buf * (A_BYTES + B_BYTES);
      char *B = A + A_BYTES;
      const int c0 = l_chunk * KE;
      const unsigned shift = (unsigned)((((l_dx * p.Y + l_dy) * p.Z + l_dz) * p.Cin + c0) * 2);
      const unsigned wshift = (unsigned)(((long long)l_tap * w_tap_stride + c0) * 2);
#pragma unroll
      for (int i = 0; i < A_RPT; ++i)
        lds_dma16(xr, A + (wave_u * (64 / PPR) + RSTEP * i) * KB, ((a_mask[i] >> l_tap) & 1u) ? a_voff[i] + shift : kOOB);
#pragma unroll
      for (int i = 0; i < B_RPT; ++i) lds_dma16(wr, B + (wave_u * (64 / PPR) + RSTEP * i) * KB, b_voff[i] + wshift);
      if (++l_chunk == cpt) {
        l_chunk = 0;
        ++l_tap;
        if (++l_dz > 1) { l_dz = -1; if (++l_dy > 1) { l_dy = -1; ++l_dx; } }
      }
    };
    // 3-stage pipeline: the DMA of step k+2 is issued while the consumers multiply step k; the wait before each barrier only
    // covers step k+1 (s_waitcnt vmcnt(12) leaves the 12 newest pieces in flight), so a tile has two full steps to land.
    issue(0);
    if (nk > 1) { issue(1); asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    int nb = 2;
    for (int ks = 0; ks < nk; ++ks) {
      if (ks + 2 < nk) {
        issue(nb);
        nb = nb == 2 ? 0 : nb + 1;
        asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    return;
  }

  // ---- consumers
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fk = lane >> 5;
  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragment addresses inside a buffer (loop invariant); sub-step s toggles the 16-byte slot by XOR
  int a_row[TM], b_row[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_row[i] = (wm * TM + i) * 32 + fr;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_row[j] = (wn * TN + j) * 32 + fr;
  auto load_frags = [&](const char *A, const char *B, int s, f4 (&af)[TM], f4 (&bfv)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) bfv[j] = *reinterpret_cast<const f4 *>(B + b_row[j] * KB + (((s * 2 + fk) ^ ((b_row[j] >> 1) & (PPR - 1))) << 4));
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f4 *>(A + a_row[i] * KB + (((s * 2 + fk) ^ ((a_row[i] >> 1) & (PPR - 1))) << 4));
  };
  // register double-buffered fragments: the ds_reads of sub-step s+1 are in flight under the 8 MFMAs of sub-step s
  auto compute = [&](int buf) {
    const char *A = lds + buf * (A_BYTES + B_BYTES);
    const char *B = A + A_BYTES;
    f4 af[2][TM], bfv[2][TN];
    load_frags(A, B, 0, af[0], bfv[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
    for (int s = 0; s < KB / 32; ++s) {
      if (s + 1 < KB / 32) load_frags(A, B, s + 1, af[(s + 1) & 1], bfv[(s + 1) & 1]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[s & 1][i], bfv[s & 1][j]);
      // pin the interleave: one ds_read of sub-step s+1 behind each of the first six MFMAs of sub-step s
      if (s + 1 < KB / 32) {
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
    }
  };
  asm volatile("s_barrier" ::: "memory");
  int cb = 0;
#pragma unroll 1
  for (int ks = 0; ks < nk; ++ks) {
    compute(cb);
    cb = cb == 2 ? 0 : cb + 1;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  const bool has_bias = (p.flags & NRPN_CONV_BIAS) && p.bias;
  const bool relu = p.flags & NRPN_CONV_RELU;
  int elane = lane;
  asm volatile("" : "+v"(elane));     // opaque copy: keeps the 128 output addresses from being hoisted above the K loop (spills)
  const int efr = elane & 31;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + efr;
    if (col >= p.Cout) continue;
    const float bv = has_bias ? p.bias[col] : 0.f;
    const float sv = p.scale ? p.scale[col] : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long v = m0 + (wm * TM + i) * 32 + frag_row(r, elane);
        if (v < p.M) {
          float o = acc[i][j][r] * sv + bv;
          if (relu) o = fmaxf(o, 0.f);
          if (p.mask && !(elem<T>::ld(reinterpret_cast<const T *>(p.mask) + v * p.Cout + col) > 0.f)) o = 0.f;
          if (OUTF32) reinterpret_cast<float *>(p.y)[v * p.Cout + col] = o;
          else elem<T>::st(reinterpret_cast<T *>(p.y) + v * p.Cout + col, o);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 tile, 8 waves (2 x 4, each 128x64), every wave loads and multiplies.  Per MFMA this halves both the bytes pulled
// through the CU's vector-memory path (64 KB per 2048 MFMA cycles: the 128x128 tile needs 32 KB per 512 and saturates the
// 64 B/clk L1 path) and the number of LDS-DMA issues (8 pieces per 32 MFMAs per wave).  Two LDS buffers of 64 KB.
// Used when Cout >= 256 and the 256x256 tiling still yields ~one workgroup per CU.
// ---------------------------------------------------------------------------------------------------------------------
// DBG (tools-only instantiations, selected by NRPN_CONV_DEBUG_* in the flag word; the production instantiation is DBG = 0 and carries
// none of these branches): bit 0 = every tap reads the centre voxel ("ideal memory"), bit 1 = no per-K-step barrier / DMA drain.
// RESULTS ARE WRONG with DBG != 0 (timing diagnosis, tools/diag_big_conv.py).
// ROWS: row-list form (tile row v = voxel p.rows[2 v], tap word p.rows[2 v + 1]; outputs and the ReLU mask at that voxel), cf. conv_igemm_kernel.
template <bool OUTF32, bool STAG = false, int DBG = 0, bool ROWS = false>
__global__ void __launch_bounds__(512, 1) conv_igemm_big_kernel(const ConvArgs p) {
  typedef bf16s T;
  constexpr int BM = 256, BN = 256, KB = 128, KE = 64, PPR = 8, RSTEP = 64, TM = 4, TN = 2;
  constexpr int A_RPT = BM / RSTEP, B_RPT = BN / RSTEP;    // 4 + 4 DMA pieces per lane per K-step
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned ntiles = (p.Cout + BN - 1) / BN;
  const unsigned tiles = (unsigned)((p.M + BM - 1) / BM) * ntiles;
  // Three tilings of the launch: every tile whole (ksplit <= 1), every tile on `ksplit` K slices, or -- tail_ks > 1 -- the M tiles from
  // tail_tile0 on (the last, partial round of a launch that needs a little more than a whole number of rounds of the chip) on tail_ks
  // K slices while the full rounds run whole: a 323-tile launch (eval at 200 x 200 x 130) then costs 1 + 1/3 rounds instead of 2.
  unsigned zsplit, tile;
  int my_ks = p.ksplit > 1 ? p.ksplit : 1;
  long long part_row0 = 0;                                         // first row of the partials' row range in ws
  if (p.tail_ks > 1) {
    const unsigned full = (unsigned)p.tail_tile0 * ntiles;
    if (blockIdx.x < full) {
      zsplit = 0; my_ks = 1;
      tile = xcd_remap(blockIdx.x, full);
    } else {
      const unsigned t = blockIdx.x - full, ntail = tiles - full;
      zsplit = t / ntail; my_ks = p.tail_ks;
      tile = full + t % ntail;
      part_row0 = (long long)p.tail_tile0 * BM;
    }
  } else {
    zsplit = blockIdx.x / tiles;                                   // K slice (0 unless ksplit > 1)
    tile = xcd_remap(blockIdx.x - zsplit * tiles, tiles);
  }
  const long long m0 = (long long)(tile / ntiles) * BM;
  const int n0 = (int)(tile % ntiles) * BN;
  const int cpt = p.Cin / KE;
  int ks_begin = 0, nk = p.taps * cpt;
  if (my_ks > 1) {
    const int per = (nk + my_ks - 1) / my_ks;
    ks_begin = zsplit * per;
    nk = min(nk, ks_begin + per);        // this workgroup runs K-steps [ks_begin, nk)
  }

  const int lr = tid / PPR, ls = tid % PPR;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
  unsigned a_voff[A_RPT], a_mask[A_RPT], b_voff[B_RPT];
  int a_yz[A_RPT], a_zs[A_RPT];            // byte strides of one x / one y step in this row's own grid
#pragma unroll
  for (int i = 0; i < A_RPT; ++i) {
    const int r = lr + RSTEP * i;
    const long long v = m0 + r;
    const bool ok = v < p.M;
    long long vv = ok ? v : 0;
    int ox, oy, oz, gX = p.X, gY = p.Y, gZ = p.Z;
    unsigned row_word = 0;
    if (ROWS) {
      if (ok) { row_word = p.rows[2 * vv + 1]; vv = p.rows[2 * vv]; }
      const int seg = (int)(row_word >> 27);
#pragma unroll
      for (int q = 0; q < kMaxSeg; ++q)
        if (q == seg && q < p.segs.n) { gY = p.segs.Y[q]; gZ = p.segs.Z[q]; }
      ox = oy = oz = 0;
    } else if (p.segs.n == 0) {    // classic layout: multiply-shift divisions (conv_common.cuh FastDiv)
      const unsigned u = (unsigned)vv, sc = fastdiv(u, p.dvs), local = u - sc * (unsigned)(p.X * p.Y * p.Z), t1 = fastdiv(local, p.dvz);
      oz = (int)(local - t1 * (unsigned)p.Z);
      ox = (int)fastdiv(t1, p.dvy);
      oy = (int)(t1 - (unsigned)ox * (unsigned)p.Y);
    } else {
      locate_voxel(p.segs, vv, p.X, p.Y, p.Z, ox, oy, oz, gX, gY, gZ);
    }
    a_yz[i] = gY * gZ * p.Cin * 2;
    a_zs[i] = gZ * p.Cin * 2;
    unsigned m = 0;
    if (ROWS) {
      m = p.taps == 27 ? (row_word & 0x7FFFFFFu) : (ok ? 1u : 0u);
    } else if (ok) {
      m = p.taps == 27 ? tap_mask27(ox, oy, oz, gX, gY, gZ) : 1u;
    }
    a_mask[i] = m;
    a_voff[i] = (unsigned)(vv * p.Cin * 2) + ((ls ^ ((r >> 1) & (PPR - 1))) << 4);
  }
#pragma unroll
  for (int i = 0; i < B_RPT; ++i) {
    const int r = lr + RSTEP * i, row = n0 + r;
    b_voff[i] = row < p.wrows ? (unsigned)((long long)row * p.Cin * 2) + ((ls ^ ((r >> 1) & (PPR - 1))) << 4) : kOOB;
  }
  int l_chunk = ks_begin / p.taps, l_tap = ks_begin - l_chunk * p.taps, l_dx = 0, l_dy = 0, l_dz = 0;
  if (p.taps == 27) { l_dx = l_tap / 9 - 1; l_dy = (l_tap / 3) % 3 - 1; l_dz = l_tap % 3 - 1; }
  const long long w_tap_stride = (long long)p.wrows * p.Cin;
  constexpr bool dbg_alias = (DBG & 1) != 0, dbg_nosync = (DBG & 2) != 0;
  // branch-free: past the last K-step (`live` false) every lane reads out of range, i.e. deposits zeros in the idle buffer
  auto issue = [&](int buf, bool live) {
    char *A = lds + buf * (A_BYTES + B_BYTES);
    char *B = A + A_BYTES;
    const int c0 = l_chunk * KE;
    const int zshift = (l_dz * p.Cin + c0) * 2;
    const unsigned wshift = (unsigned)(((long long)l_tap * w_tap_stride + c0) * 2);
    const unsigned tapbit = live ? (1u << l_tap) : 0u;
    if constexpr (dbg_alias) {      // timing diagnosis only: every tap reads the centre voxel (27x reuse of one activation tile: "ideal memory")
#pragma unroll
      for (int i = 0; i < A_RPT; ++i) lds_dma16(xr, A + (wave_u * (64 / PPR) + RSTEP * i) * KB, live ? a_voff[i] + (unsigned)(c0 * 2) : kOOB);
    } else {
#pragma unroll
    for (int i = 0; i < A_RPT; ++i)
      lds_dma16(xr, A + (wave_u * (64 / PPR) + RSTEP * i) * KB,
                (a_mask[i] & tapbit) ? a_voff[i] + (unsigned)(l_dx * a_yz[i] + l_dy * a_zs[i] + zshift) : kOOB);
    }
#pragma unroll
    for (int i = 0; i < B_RPT; ++i) lds_dma16(wr, B + (wave_u * (64 / PPR) + RSTEP * i) * KB, live ? b_voff[i] + wshift : kOOB);
    // K order: 64-channel chunk OUTER, tap INNER.  All workgroups of an XCD then sweep the 27 shifted views of one 128-byte
    // channel slice of their voxel slab (~1.4 MB per XCD) before moving to the next slice, which fits the 4 MB L2; with the
    // tap-outer order the full 512-byte rows (5.8 MB per XCD) were streamed 27 times through it (85 % hit rate).
    if (++l_tap == p.taps) {
      l_tap = 0;
      ++l_chunk;
      if (p.taps == 27) { l_dx = -1; l_dy = -1; l_dz = -1; }
    } else if (p.taps == 27) {
      if (++l_dz > 1) { l_dz = -1; if (++l_dy > 1) { l_dy = -1; ++l_dx; } }
    }
  };

  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 31, fk = lane >> 5;
  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int a_row[TM], b_row[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_row[i] = (wm * TM + i) * 32 + fr;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_row[j] = (wn * TN + j) * 32 + fr;
  auto load_frags = [&](const char *A, const char *B, int s, f4 (&af)[TM], f4 (&bfv)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) bfv[j] = *reinterpret_cast<const f4 *>(B + b_row[j] * KB + (((s * 2 + fk) ^ ((b_row[j] >> 1) & (PPR - 1))) << 4));
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f4 *>(A + a_row[i] * KB + (((s * 2 + fk) ^ ((a_row[i] >> 1) & (PPR - 1))) << 4));
  };
  // one K-step: the 8 LDS-DMA pieces of the NEXT step are issued one behind each MFMA of sub-step 0 (an LDS-DMA issue
  // costs ~60 cycles among bare MFMAs but >100 in a burst), the ds_reads of sub-step s+1 behind the MFMAs of sub-step s
  // one K-step: the 8 LDS-DMA pieces of the NEXT step are issued one behind each MFMA of sub-step 0 (an LDS-DMA issue
  // costs ~60 cycles among bare MFMAs but >100 in a burst), the ds_reads of sub-step s+1 behind the MFMAs of sub-step s.
  // STAG (experiment): the two waves of a SIMD leave the barrier together and would both sit in their DMA-issue phase at once;
  // waves 4-7 issue theirs as one burst before sub-step 2 instead (a wave-uniform branch cannot be interleaved by the scheduler).
  const bool late_issue = STAG && wave_u >= 4;
  auto compute = [&](int buf, bool next_live) {
    const char *A = lds + buf * (A_BYTES + B_BYTES);
    const char *B = A + A_BYTES;
    f4 af[2][TM], bfv[2][TN];
    load_frags(A, B, 0, af[0], bfv[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
    if (!STAG) issue(buf ^ 1, next_live);
#pragma unroll
    for (int s = 0; s < KB / 32; ++s) {
      if (s + 1 < KB / 32) load_frags(A, B, s + 1, af[(s + 1) & 1], bfv[(s + 1) & 1]);
      if (STAG && (s == 0 || s == 2)) {
        if ((s == 2) == late_issue) issue(buf ^ 1, next_live);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[s & 1][i], bfv[s & 1][j]);
      if (!STAG && s == 0) {
#pragma unroll
        for (int q = 0; q < TM * TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          if (q < TM + TN) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else if (s + 1 < KB / 32) {
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      }
    }
  };

  issue(0, ks_begin < nk);
  __syncthreads();
  if constexpr (STAG) {
    // Rotated pipeline: the barrier that hands over the next buffer sits BEFORE the MFMAs of the last sub-step instead of after them.
    // The fragments of sub-step 3 are in registers by then, so its 8 MFMAs run while the ds_reads of the next K-step's sub-step 0 are in
    // flight -- in the plain loop every wave of the workgroup left the barrier with nothing to multiply until those reads returned.
    // Same MFMA order per accumulator as the plain loop: bit-identical results.  DMA issue: waves 0-3 in sub-step 0, waves 4-7 in
    // sub-step 1 (the two waves of a SIMD are then not both in their DMA-issue phase); both land before the barrier after sub-step 2.
    f4 af[2][TM], bfv[2][TN];
    load_frags(lds, lds + A_BYTES, 0, af[0], bfv[0]);
#pragma unroll 1
    for (int ks = ks_begin; ks < nk; ++ks) {
      const int buf = (ks - ks_begin) & 1;
      const char *A = lds + buf * (A_BYTES + B_BYTES);
      const char *B = A + A_BYTES;
      const char *An = lds + (buf ^ 1) * (A_BYTES + B_BYTES);
      const bool next_live = ks + 1 < nk;
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        if (sub < 3) {
          load_frags(A, B, sub + 1, af[(sub + 1) & 1], bfv[(sub + 1) & 1]);
        } else {
          if constexpr (!dbg_nosync) __syncthreads();
          load_frags(An, An + A_BYTES, 0, af[0], bfv[0]);
        }
        if (sub < 2 && (sub == 1) == late_issue) issue(buf ^ 1, next_live);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[sub & 1][i], bfv[sub & 1][j]);
#pragma unroll
        for (int q = 0; q < TM + TN; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
      }
    }
  } else {
#pragma unroll 1
    for (int ks = ks_begin; ks < nk; ++ks) {
      compute((ks - ks_begin) & 1, ks + 1 < nk);
      if constexpr (!dbg_nosync) __syncthreads();
    }
  }

  const bool has_bias = (p.flags & NRPN_CONV_BIAS) && p.bias;
  const bool relu = p.flags & NRPN_CONV_RELU;
  int elane = lane;
  asm volatile("" : "+v"(elane));
  const int efr = elane & 31;
  if (my_ks > 1) {     // fp32 partial of this K slice, plain stores; bias / ReLU / cast happen in splitk_epilogue_kernel
    float *wsz = p.ws + (long long)zsplit * (p.M - part_row0) * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + efr;
      if (col >= p.Cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long v = m0 + (wm * TM + i) * 32 + frag_row(r, elane);
          if (v < p.M) wsz[(v - part_row0) * p.Cout + col] = acc[i][j][r];
        }
    }
    return;
  }
  if (!OUTF32 && (p.Cout & 7) == 0) {
    // bf16 output: the C fragment of a lane is one 2-byte element per row -- 128 two-byte global stores (and as many two-byte mask
    // loads) per wave.  Stage each wave's 128x64 block through LDS instead (free after the K loop: every wave has passed the last
    // barrier), 64 rows at a time with a 144-byte row pitch (rows 4 apart land on different banks), and move it out as whole
    // 16-byte pieces: 8 lanes cover one 128-byte row segment, the optional ReLU mask comes in with the same 16-byte loads.
    constexpr int PITCH = 144;
    char *stage = lds + wave * (64 * PITCH);
    const T *maskp = reinterpret_cast<const T *>(p.mask);
    T *yp = reinterpret_cast<T *>(p.y);
    const bool full_tile = m0 + BM <= p.M;
    float ssum[TN] = {0.f, 0.f}, qsum[TN] = {0.f, 0.f};
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
            float o = acc[i][j][r] * sv + bv;
            if (relu) o = fmaxf(o, 0.f);
            const bf16s ob = f32_to_bf16_bits(o);
            *reinterpret_cast<bf16s *>(stage + (ii * 32 + frag_row(r, elane)) * PITCH + (j * 32 + efr) * 2) = ob;
            if (p.stats && (full_tile || m0 + (wm * TM + i) * 32 + frag_row(r, elane) < p.M)) {
              const float of = bf16_bits_to_f32(ob);
              ssum[j] += of;
              qsum[j] += of * of;
            }
          }
        }
      }
      // same wave wrote and reads: only the LDS counter has to drain
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int pc = elane + 64 * q;                  // 512 pieces: 64 rows x 8 pieces of 16 bytes
        const int row = pc >> 3, seg = pc & 7;
        const long long vt = m0 + wm * 128 + half * 64 + row;
        const int col = n0 + wn * 64 + seg * 8;
        if (vt < p.M && col < p.Cout) {
          const long long v = ROWS ? (long long)p.rows[2 * vt] : vt;
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of this half before the next half overwrites the stage
    }
    if (p.stats) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        store_col_stats(p.stats, (m0 / BM) * 2 + wm, p.Cout, n0 + (wn * TN + j) * 32 + efr, ssum[j], qsum[j], elane);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + efr;
    if (col >= p.Cout) continue;
    const float bv = has_bias ? p.bias[col] : 0.f;
    const float sv = p.scale ? p.scale[col] : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long vt = m0 + (wm * TM + i) * 32 + frag_row(r, elane);
        if (vt < p.M) {
          const long long v = ROWS ? (long long)p.rows[2 * vt] : vt;
          float o = acc[i][j][r] * sv + bv;
          if (relu) o = fmaxf(o, 0.f);
          if (p.mask && !(elem<T>::ld(reinterpret_cast<const T *>(p.mask) + v * p.Cout + col) > 0.f)) o = 0.f;
          if (OUTF32) reinterpret_cast<float *>(p.y)[v * p.Cout + col] = o;
          else elem<T>::st(reinterpret_cast<T *>(p.y) + v * p.Cout + col, o);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256x256 tile on FOUR waves (2 x 2, each 128x128 = 16 accumulators = 256 accumulator registers; one wave per SIMD, 512 registers).
// Per MFMA this is 0.5 ds_read_b128 instead of the 0.75 of the 8-wave kernel (a third less LDS read traffic and energy -- the part is
// power-limited on real data) at the same global -> LDS traffic; the price is that nothing hides an issue stall of the single wave
// of a SIMD, so the 16 LDS-DMA pieces per wave per K-step are spread one behind every other MFMA pair of sub-steps 0..2 and the
// ds_reads of sub-step s+1 sit one behind every second MFMA of sub-step s.  Same rotated K-step (barrier before the MFMAs of the
// last sub-step) and the same per-accumulator MFMA order as conv_igemm_big_kernel: results are bit-identical to it.
// Classic layout only (no ragged voxel lists).  Selected per call (nrpn_conv_opts.tile = NRPN_TILE_256X256_W4) or by the plan.
// ---------------------------------------------------------------------------------------------------------------------
template <bool OUTF32>
__global__ void __launch_bounds__(256, 1) conv_igemm_big4_kernel(const ConvArgs p) {
  typedef bf16s T;
  constexpr int BM = 256, BN = 256, KB = 128, KE = 64, PPR = 8, RSTEP = 32, TM = 4, TN = 4;
  constexpr int A_RPT = BM / RSTEP, B_RPT = BN / RSTEP;    // 8 + 8 DMA pieces per lane per K-step
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB;
  extern __shared__ __attribute__((aligned(16))) char lds[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned ntiles = (p.Cout + BN - 1) / BN;
  const unsigned tiles = (unsigned)((p.M + BM - 1) / BM) * ntiles;
  const unsigned zsplit = blockIdx.x / tiles;                      // K slice (0 unless ksplit > 1)
  const unsigned tile = xcd_remap(blockIdx.x - zsplit * tiles, tiles);
  const long long m0 = (long long)(tile / ntiles) * BM;
  const int n0 = (int)(tile % ntiles) * BN;
  const int cpt = p.Cin / KE;
  int ks_begin = 0, nk = p.taps * cpt;
  if (p.ksplit > 1) {
    const int per = (nk + p.ksplit - 1) / p.ksplit;
    ks_begin = zsplit * per;
    nk = min(nk, ks_begin + per);
  }

  const int lr = tid / PPR, ls = tid % PPR;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
  unsigned a_voff[A_RPT], a_mask[A_RPT], b_voff[B_RPT];
  const int yz_b = p.Y * p.Z * p.Cin * 2, zs_b = p.Z * p.Cin * 2;      // byte strides of one x / one y step
#pragma unroll
  for (int i = 0; i < A_RPT; ++i) {
    const int r = lr + RSTEP * i;
    const long long v = m0 + r;
    const bool ok = v < p.M;
    const long long vv = ok ? v : 0;
    const int oz = (int)(vv % p.Z);
    const long long t1 = vv / p.Z;
    const int oy = (int)(t1 % p.Y);
    const int ox = (int)((t1 / p.Y) % p.X);
    unsigned m = 0;
    if (ok) {
      if (p.taps == 27) {
#pragma unroll
        for (int t = 0; t < 27; ++t) {
          const int dx = t / 9 - 1, dy = (t / 3) % 3 - 1, dz = t % 3 - 1;
          const bool in = (unsigned)(ox + dx) < (unsigned)p.X && (unsigned)(oy + dy) < (unsigned)p.Y && (unsigned)(oz + dz) < (unsigned)p.Z;
          m |= in ? (1u << t) : 0u;
        }
      } else {
        m = 1u;
      }
    }
    a_mask[i] = ~m | (1u << 27);      // INVERTED: bit t set = tap t reads zeros for this row; bit 27 = "past the last K-step"
    a_voff[i] = (unsigned)(vv * p.Cin * 2) + ((ls ^ ((r >> 1) & (PPR - 1))) << 4);
  }
#pragma unroll
  for (int i = 0; i < B_RPT; ++i) {
    const int r = lr + RSTEP * i, row = n0 + r;
    b_voff[i] = row < p.wrows ? (unsigned)((long long)row * p.Cin * 2) + ((ls ^ ((r >> 1) & (PPR - 1))) << 4) : kOOB;
  }
  // K order as in conv_igemm_big_kernel: 64-channel chunk OUTER, tap INNER (one channel slice of the XCD's voxel slab stays in L2)
  int l_chunk = ks_begin / p.taps, l_tap = ks_begin - l_chunk * p.taps, l_dx = 0, l_dy = 0, l_dz = 0;
  if (p.taps == 27) { l_dx = l_tap / 9 - 1; l_dy = (l_tap / 3) % 3 - 1; l_dz = l_tap % 3 - 1; }
  const long long w_tap_stride = (long long)p.wrows * p.Cin;
  // one DMA piece of the K-step described by (l_tap, l_chunk, l_d*): pieces 0..7 = A rows, 8..15 = B rows.  Branch-free and three
  // VALU per A piece: bit = bfe(inverted mask, tap), offset = (row offset + tap shift) | bit << 31 -- an offset >= 2^31 is beyond
  // every descriptor (tensors stay below 2 GiB), so the lane deposits zeros; `live` false selects bit 27 (always set).
  const bool is27 = p.taps == 27;
  auto piece = [&](int buf, bool live, int idx) {
    char *A = lds + buf * (A_BYTES + B_BYTES);
    char *B = A + A_BYTES;
    const int c0 = l_chunk * KE;
    if (idx < A_RPT) {
      const unsigned sel = live ? (unsigned)l_tap : 27u;
      const unsigned shift = (unsigned)(l_dx * yz_b + l_dy * zs_b + (l_dz * p.Cin + c0) * 2);
      const unsigned bit = __builtin_amdgcn_ubfe(a_mask[idx], sel, 1u);
      lds_dma16(xr, A + (wave_u * (64 / PPR) + RSTEP * idx) * KB, (bit << 31) | (a_voff[idx] + shift));
    } else {
      const unsigned wshift = live ? (unsigned)(((long long)l_tap * w_tap_stride + c0) * 2) : kOOB;
      lds_dma16(wr, B + (wave_u * (64 / PPR) + RSTEP * (idx - A_RPT)) * KB, b_voff[idx - A_RPT] + wshift);
    }
  };
  auto advance = [&]() {      // scalar selects only: the K-step body must stay ONE basic block for the instruction interleave below
    const int nt = l_tap + 1;
    const bool wrap = nt == p.taps;
    l_tap = wrap ? 0 : nt;
    l_chunk += wrap ? 1 : 0;
    const int nz = l_dz + 1;
    const bool cz = nz > 1;
    const int ny = l_dy + (cz ? 1 : 0);
    const bool cy = ny > 1;
    const int nx = l_dx + (cy ? 1 : 0);
    l_dz = !is27 ? 0 : (wrap ? -1 : (cz ? -1 : nz));
    l_dy = !is27 ? 0 : (wrap ? -1 : (cy ? -1 : ny));
    l_dx = !is27 ? 0 : (wrap ? -1 : nx);
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fk = lane >> 5;
  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int a_row[TM], b_row[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_row[i] = (wm * TM + i) * 32 + fr;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_row[j] = (wn * TN + j) * 32 + fr;
  auto load_frags = [&](const char *A, const char *B, int s, f4 (&af)[TM], f4 (&bfv)[TN]) {
#pragma unroll
    for (int j = 0; j < TN; ++j) bfv[j] = *reinterpret_cast<const f4 *>(B + b_row[j] * KB + (((s * 2 + fk) ^ ((b_row[j] >> 1) & (PPR - 1))) << 4));
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f4 *>(A + a_row[i] * KB + (((s * 2 + fk) ^ ((a_row[i] >> 1) & (PPR - 1))) << 4));
  };

#pragma unroll
  for (int q = 0; q < A_RPT + B_RPT; ++q) piece(0, ks_begin < nk, q);
  advance();
  __syncthreads();
  f4 af[2][TM], bfv[2][TN];
  load_frags(lds, lds + A_BYTES, 0, af[0], bfv[0]);
  constexpr int kPieces[4] = {8, 8, 0, 0};      // DMA pieces issued in sub-steps 0..3: a whole sub-step (512 MFMA cycles) to land before the barrier
#pragma unroll 1
  for (int ks = ks_begin; ks < nk; ++ks) {
    const int buf = (ks - ks_begin) & 1;
    const char *A = lds + buf * (A_BYTES + B_BYTES);
    const char *B = A + A_BYTES;
    const char *An = lds + (buf ^ 1) * (A_BYTES + B_BYTES);
    const bool next_live = ks + 1 < nk;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      if (sub < 3) {
        load_frags(A, B, sub + 1, af[(sub + 1) & 1], bfv[(sub + 1) & 1]);
      } else {
        __syncthreads();
        load_frags(An, An + A_BYTES, 0, af[0], bfv[0]);
      }
      constexpr int kFirst[4] = {0, 8, 16, 16};
#pragma unroll
      for (int q = 0; q < kPieces[sub]; ++q) piece(buf ^ 1, next_live, kFirst[sub] + q);
      if (sub == 1) advance();
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[sub & 1][i], bfv[sub & 1][j]);
      // pin the interleave: per pair of MFMAs one ds_read of the next sub-step, and (while pieces remain) one LDS-DMA piece
#pragma unroll
      for (int q = 0; q < TM + TN; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (q < kPieces[sub]) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
  }

  const bool has_bias = (p.flags & NRPN_CONV_BIAS) && p.bias;
  const bool relu = p.flags & NRPN_CONV_RELU;
  int elane = lane;
  asm volatile("" : "+v"(elane));
  const int efr = elane & 31;
  if (p.ksplit > 1) {     // fp32 partial of this K slice, plain stores; bias / ReLU / cast happen in splitk_epilogue_kernel
    float *wsz = p.ws + (long long)zsplit * p.M * p.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + efr;
      if (col >= p.Cout) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long v = m0 + (wm * TM + i) * 32 + frag_row(r, elane);
          if (v < p.M) wsz[v * p.Cout + col] = acc[i][j][r];
        }
    }
    return;
  }
  if (!OUTF32 && (p.Cout & 7) == 0) {
    // staged bf16 epilogue (see conv_igemm_big_kernel): 32 rows x 128 columns of the wave's block at a time through LDS (272-byte row
    // pitch: rows 4 apart land on different banks), out as whole 16-byte pieces, 16 lanes per 256-byte row segment
    constexpr int PITCH = 272;
    char *stage = lds + wave * (32 * PITCH);
    const T *maskp = reinterpret_cast<const T *>(p.mask);
    T *yp = reinterpret_cast<T *>(p.y);
    const bool full_tile = m0 + BM <= p.M;
    float ssum[TN], qsum[TN], bv[TN], sv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + (wn * TN + j) * 32 + efr;
      ssum[j] = 0.f; qsum[j] = 0.f;
      bv[j] = (has_bias && col < p.Cout) ? p.bias[col] : 0.f;
      sv[j] = (p.scale && col < p.Cout) ? p.scale[col] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float o = acc[i][j][r] * sv[j] + bv[j];
          if (relu) o = fmaxf(o, 0.f);
          const bf16s ob = f32_to_bf16_bits(o);
          *reinterpret_cast<bf16s *>(stage + frag_row(r, elane) * PITCH + (j * 32 + efr) * 2) = ob;
          if (p.stats && (full_tile || m0 + (wm * TM + i) * 32 + frag_row(r, elane) < p.M)) {
            const float of = bf16_bits_to_f32(ob);
            ssum[j] += of;
            qsum[j] += of * of;
          }
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int pc = elane + 64 * q;                  // 512 pieces: 32 rows x 16 pieces of 16 bytes
        const int row = pc >> 4, seg = pc & 15;
        const long long v = m0 + (wm * TM + i) * 32 + row;
        const int col = n0 + wn * 128 + seg * 8;
        if (v < p.M && col < p.Cout) {
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of this block before the next one overwrites the stage
    }
    if (p.stats) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
        store_col_stats(p.stats, (m0 / BM) * 2 + wm, p.Cout, n0 + (wn * TN + j) * 32 + efr, ssum[j], qsum[j], elane);
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + (wn * TN + j) * 32 + efr;
    if (col >= p.Cout) continue;
    const float bv = has_bias ? p.bias[col] : 0.f;
    const float sv = p.scale ? p.scale[col] : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long v = m0 + (wm * TM + i) * 32 + frag_row(r, elane);
        if (v < p.M) {
          float o = acc[i][j][r] * sv + bv;
          if (relu) o = fmaxf(o, 0.f);
          if (p.mask && !(elem<T>::ld(reinterpret_cast<const T *>(p.mask) + v * p.Cout + col) > 0.f)) o = 0.f;
          if (OUTF32) reinterpret_cast<float *>(p.y)[v * p.Cout + col] = o;
          else elem<T>::st(reinterpret_cast<T *>(p.y) + v * p.Cout + col, o);
        }
      }
    }
  }
}

// conv_halo.hip: the "halo" form of the 3x3x3 implicit GEMM (4 x 8 x 8 voxel blocks, input halo staged once per channel chunk)
namespace hk { constexpr int TX = 4, TY = 8, TZ = 8; }
int nrpn_launch_conv_halo(const ConvArgs &a, unsigned workgroups, hipStream_t st, int variant);

template <typename T, bool OUTF32>
__global__ void splitk_epilogue_kernel(const float *__restrict__ ws, const float *__restrict__ bias, void *__restrict__ y, long long total,
                                       int cout, int relu, int nslices, const T *__restrict__ mask, const float *__restrict__ scale) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float o = ws[i];
    for (int z = 1; z < nslices; ++z) o += ws[z * total + i];
    if (scale) o *= scale[i % cout];
    o += bias ? bias[i % cout] : 0.f;
    if (relu) o = fmaxf(o, 0.f);
    if (mask && !(elem<T>::ld(mask + i) > 0.f)) o = 0.f;
    if (OUTF32) reinterpret_cast<float *>(y)[i] = o;
    else elem<T>::st(reinterpret_cast<T *>(y) + i, o);
  }
}

// the same for the row-list form: partial row r belongs to voxel rows[2 r]; bias / ReLU / ReLU mask / store at that voxel's row of y
template <typename T, bool OUTF32>
__global__ void rows_epilogue_kernel(const float *__restrict__ ws, const float *__restrict__ bias, void *__restrict__ y, long long total,
                                     int cout, int relu, int nslices, const T *__restrict__ mask, const unsigned *__restrict__ rows) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float o = ws[i];
    for (int z = 1; z < nslices; ++z) o += ws[z * total + i];
    const long long r = i / cout;
    const int col = (int)(i - r * cout);
    const long long dst = (long long)rows[2 * r] * cout + col;
    o += bias ? bias[col] : 0.f;
    if (relu) o = fmaxf(o, 0.f);
    if (mask && !(elem<T>::ld(mask + dst) > 0.f)) o = 0.f;
    if (OUTF32) reinterpret_cast<float *>(y)[dst] = o;
    else elem<T>::st(reinterpret_cast<T *>(y) + dst, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Kernel selection.  Every decision below is a pure function of (shape, Knobs): a call resolves its Knobs from the process-wide
// defaults (developer switches, tools only) overridden by the caller's nrpn_conv_opts, so two threads with different plans never
// touch shared state while launching.
// ---------------------------------------------------------------------------------------------------------------------
struct Knobs {
  int glds;        // 1: LDS-DMA loads (buffer_load ... lds), 0: register-staged loads
  int bm;          // tile selector: 0 auto, 128, 256 (wave-specialised 256x128), 512 (256x256, 8 waves), 1024 (256x256, 4 waves), 2048 (halo form)
  int stagger;     // 256x256 8-wave kernel: rotated K-step + staggered DMA issue (default) or the plain loop
  int big_split;   // mid-size grids run the 256x256 kernel on K slices
  int kb;          // K-step bytes of the k1/k3 kernels (64 or 128)
  int dbg;         // tools only: NRPN_CONV_DEBUG_* bits
  int halo_auto;   // 1 (default): the halo form is chosen automatically where it applies (40^3-class grids), 0: only on request (tile 2048)
  int halo_xp;     // halo form: taps paired across chunk boundaries (54 full K-steps per four chunks; Cin % 128 == 0) instead of 14 per chunk
};
static std::atomic<int> g_conv_glds{1}, g_conv_bm{0}, g_conv_stagger{1}, g_conv_big_split{1}, g_conv_kb{128}, g_conv_halo_auto{1}, g_conv_halo_xp{NRPN_HALO_PAIRING_DEFAULT};
static Knobs resolve_knobs(const nrpn_conv_opts *o, int flags = 0) {
  Knobs k{g_conv_glds.load(std::memory_order_relaxed), g_conv_bm.load(std::memory_order_relaxed), g_conv_stagger.load(std::memory_order_relaxed),
          g_conv_big_split.load(std::memory_order_relaxed), g_conv_kb.load(std::memory_order_relaxed),
          flags & (NRPN_CONV_DEBUG_ALIAS_TAPS | NRPN_CONV_DEBUG_NO_SYNC | NRPN_CONV_DEBUG_STAGGER | NRPN_CONV_DEBUG_VARIANT), g_conv_halo_auto.load(std::memory_order_relaxed),
          g_conv_halo_xp.load(std::memory_order_relaxed)};
  if (o) {
    if (o->halo_pairing == 1 || o->halo_pairing == 2) k.halo_xp = o->halo_pairing == 1 ? 1 : 0;
    if (o->tile > 0) k.bm = o->tile;
    if (o->lds_dma >= 0) k.glds = o->lds_dma ? 1 : 0;
    if (o->kstep_bytes == 64 || o->kstep_bytes == 128) k.kb = o->kstep_bytes;
    if (o->stagger >= 0) k.stagger = o->stagger ? 1 : 0;
    if (o->big_split >= 0) k.big_split = o->big_split ? 1 : 0;
    k.dbg |= o->debug & (NRPN_CONV_DEBUG_ALIAS_TAPS | NRPN_CONV_DEBUG_NO_SYNC | NRPN_CONV_DEBUG_STAGGER | NRPN_CONV_DEBUG_VARIANT);
  }
  return k;
}
extern "C" int nrpn_set_conv_lds_dma(int on) { g_conv_glds = on ? 1 : 0; return NRPN_OK; }
extern "C" int nrpn_set_conv_halo_auto(int on) { g_conv_halo_auto = on ? 1 : 0; return NRPN_OK; }
extern "C" int nrpn_set_conv_halo_pairing(int mode) { g_conv_halo_xp = (mode == 1 || mode == 2) ? mode : 0; return NRPN_OK; }
extern "C" int nrpn_set_conv_stagger(int on) { g_conv_stagger = on ? 1 : 0; return NRPN_OK; }
extern "C" int nrpn_set_conv_big_split(int on) { g_conv_big_split = on ? 1 : 0; return NRPN_OK; }
extern "C" int nrpn_set_conv_kstep_bytes(int kb) {
  if (kb != 64 && kb != 128) return nrpn_fail(NRPN_ERR_ARG, "conv k-step must be 64 or 128 bytes");
  g_conv_kb = kb;
  return NRPN_OK;
}
extern "C" int nrpn_set_conv_tile_m(int bm) {
  if (bm != 0 && bm != 128 && bm != 256 && bm != 512 && bm != 1024 && bm != 2048)
    return nrpn_fail(NRPN_ERR_ARG, "conv tile selector must be 0 (auto), 128 (128x128), 256 (wave-specialised 256x128), 512 (256x256, 8 waves), "
                                   "1024 (256x256, 4 waves) or 2048 (halo form of the 3x3x3 kernel)");
  g_conv_bm = bm;
  return NRPN_OK;
}

// split the K loop when the (M, N) tiling alone cannot fill 256 CUs (the 10^3 / 5^3 pyramid levels)
static int conv_ksplit(long long M, int cout, int cin, int taps, int elem_bytes, const Knobs &kn) {
  const int bn = cout <= 64 ? 64 : 128;
  const long long tiles = cdiv64(M, 128) * ((cout + bn - 1) / bn);
  const int kb = (kn.kb == 128 && (cin * elem_bytes) % 128 == 0) ? 128 : 64;
  const int nk = taps * (cin * elem_bytes / kb);
  if (tiles >= 128 || nk < 16) return 1;
  long long s = (384 + tiles - 1) / tiles;
  if (s > nk / 8) s = nk / 8;
  if (s > 64) s = 64;
  if (s < 2) return 1;
  const long long per = (nk + s - 1) / s;
  s = (nk + per - 1) / per;                  // no empty slice: every slice writes its whole partial
  return s < 2 ? 1 : (int)s;
}

// Mid-size grids (20^3: M = 8000) give only 32-64 tiles of 256x256: run the big kernel on K slices whose fp32 partials are
// written with plain stores to ws[z][M][Cout] and summed by the epilogue (no atomics).  Returns the slice count, 0 = not used.
static int conv_big_split(long long M, int cout, int cin, int taps, int elem_bytes, const Knobs &kn) {
  if (!kn.big_split) return 0;
  if (elem_bytes != 2 || !kn.glds || kn.kb != 128 || (kn.bm != 0 && kn.bm != 512 && kn.bm != 1024) || cout < 256 || (cin * 2) % 128 != 0) return 0;
  const long long tiles = cdiv64(M, 256) * ((cout + 255) / 256);
  // measured inside the bench (20^3 maps): 512->512 (64 tiles, 4 slices) 123 us here vs 204 us on the 128-row kernel, but 256->256
  // (32 tiles, 8 slices of 13 K-steps) 56 us here vs 49 us there -- below ~48 tiles the slices get too short to amortise the 256 KB
  // partial-tile epilogue
  if (tiles < 48 || tiles >= 200) return 0;
  const int nk = taps * (cin / 64);
  int s = (int)((256 + tiles - 1) / tiles);
  if (s > 8) s = 8;
  while (s > 1 && nk / s < 8) --s;
  return s >= 2 ? s : 0;
}

// Tail split of a 256x256-tile launch: tiles = whole rounds of the 256 CUs + a short remainder (eval at 200 x 200 x 130: 323 = 256 + 67).  The
// remainder's M tiles run on K slices so that the second round lasts 1 / ks of a round (+ a small epilogue over those rows) instead of a
// whole one.  Only single-column tilings (Cout <= 256), bf16 in / bf16 out, no fused statistics.  -> {first tail M tile, ks}; ks 0 = none.
struct TailSplit { int tile0, ks; };
static TailSplit conv_tail_split(long long M, int cout, int cin, int taps, int elem_bytes, const Knobs &kn) {
  TailSplit none{0, 0};
  if (elem_bytes != 2 || !kn.glds || kn.kb != 128 || !kn.big_split || cout > 256 || cout < 256 || (cin * 2) % 128 != 0) return none;
  const long long tiles = cdiv64(M, 256);
  const long long rem = tiles % 256;
  if (tiles <= 256 || rem == 0 || rem > 128) return none;
  const int nk = taps * (cin / 64);
  int ks = (int)(256 / rem);
  if (ks > 8) ks = 8;
  while (ks > 1 && nk / ks < 8) --ks;
  if (ks < 2) return none;
  const int per = (nk + ks - 1) / ks;
  ks = (nk + per - 1) / per;                 // no empty slice
  return ks >= 2 ? TailSplit{(int)(tiles - rem), ks} : none;
}

template <typename K>
static int launch_igemm(K kernel, dim3 grid, size_t lds, hipStream_t st, const ConvArgs &a) {
  if (lds > 64 * 1024) NRPN_LDS(kernel, (int)lds);
  hipLaunchKernelGGL(kernel, grid, dim3(256), lds, st, a);
  return NRPN_OK;
}

// Halo form (conv_halo_kernel): bf16 3x3x3 on a classic grid whose 4 x 8 x 8 blocks waste little and still fill the chip.
struct Grid { int n, gx, gy, gz; };
static long long halo_tiles(const Grid &g) {
  return (long long)g.n * ((g.gx + hk::TX - 1) / hk::TX) * ((g.gy + hk::TY - 1) / hk::TY) * ((g.gz + hk::TZ - 1) / hk::TZ);
}
static bool halo_ok(const Grid &g, int cin, int cout, int taps, int elem_bytes, bool out_f32, const Knobs &kn) {
  (void)out_f32;      // round 5: the halo kernel has an fp32-output epilogue (no mask / statistics there: the entry points reject those)
  if (g.n <= 0 || taps != 27 || elem_bytes != 2 || !kn.glds || (cin % 32) != 0 || (cout & 7) != 0 || cout < 256) return false;
  if (kn.bm != 2048 && !(kn.bm == 0 && kn.halo_auto)) return false;
  const long long tiles = halo_tiles(g) * ((cout + 255) / 256);
  const double waste = (double)halo_tiles(g) * 256.0 / ((double)g.n * g.gx * g.gy * g.gz);
  if (kn.bm == 2048) return true;                       // forced (tools / tests): any grid
  return tiles >= 200 && waste <= 1.12;
}

// the tile family a (shape, Knobs) pair selects: 0 = 128-row, 1 = 256x256 8-wave, 4 = wave-specialised 256x128, 5 = 256x256 4-wave
static int conv_tile_kind(long long M, int cout, int cin, int elem_bytes, bool sliced_big, bool ragged, const Knobs &kn) {
  const bool wide = kn.kb == 128 && (cin * elem_bytes) % 128 == 0;
  const bool can = kn.glds && wide && cout > 64 && elem_bytes == 2;
  const long long tiles_big = cdiv64(M, 256) * ((cout + 255) / 256);
  const bool huge = can && cout >= 256 && (sliced_big || kn.bm == 512 || kn.bm == 1024 || (kn.bm == 0 && tiles_big >= 200));
  if (huge) return (kn.bm == 1024 && !ragged) ? 5 : 1;
  if (can && kn.bm == 256 && !ragged) return 4;
  return 0;
}

template <typename T, int MODE>
static int launch_conv(const ConvArgs &a, bool out_f32, hipStream_t st, const Knobs &kn) {
  const int bn = (a.Cout <= 64) ? 64 : 128;
  constexpr bool kDma = (MODE == 0);
  // the 128-byte K-step needs Cin*elemsize % 128 == 0; the stem gather keeps the 64-byte step
  const bool wide = MODE == 0 && kn.kb == 128 && (a.Cin * (int)sizeof(T)) % 128 == 0;
  const int kind = (kDma && (a.ksplit <= 1 || a.slices)) ? conv_tile_kind(a.M, a.Cout, a.Cin, (int)sizeof(T), a.slices != 0, a.segs.n > 0, kn) : 0;
  const bool huge = kind == 1 || kind == 5, big = kind == 4;
  const int bm = (big || huge) ? 256 : 128;
  if (a.stats) {      // fused BatchNorm statistics exist in the staged bf16 epilogues of the 128-row and 256x256 kernels only
    const bool staged = MODE == 0 && sizeof(T) == 2 && !out_f32 && (a.Cout & 7) == 0 && a.ksplit <= 1 && !big && (huge || wide);
    if (!staged) return nrpn_fail(NRPN_ERR_ARG, "conv3d_fwd_stats: this shape does not run a kernel with fused statistics (ask nrpn_conv3d_fwd_stats_rows first)");
  }
  dim3 grid((unsigned)(cdiv64(a.M, bm) * ((a.Cout + (huge ? 256 : bn) - 1) / (huge ? 256 : bn)) * (a.ksplit > 1 ? a.ksplit : 1)));
  if (a.tail_ks > 1) {
    if (kind != 1) return nrpn_fail(NRPN_ERR_ARG, "conv3d_fwd: internal: tail split planned for a shape that does not run the 256x256 tile");
    const long long mt = cdiv64(a.M, 256), nt = (a.Cout + 255) / 256;
    grid = dim3((unsigned)(a.tail_tile0 * nt + (mt - a.tail_tile0) * nt * a.tail_ks));
  }
  int rc = 0;
#define NRPN_LC(BN_, OF_, KB_)                                                                                                    \
  do {                                                                                                                           \
    const size_t lds_ = 2 * (size_t)(128 + BN_) * KB_;                                                                           \
    if (kDma && kn.glds) rc = launch_igemm(conv_igemm_kernel<T, BN_, MODE, OF_, KB_, kDma>, grid, lds_, st, a);                  \
    else rc = launch_igemm(conv_igemm_kernel<T, BN_, MODE, OF_, KB_, false>, grid, lds_, st, a);                                 \
  } while (0)
#define NRPN_LC2(BN_, OF_) do { if (wide) NRPN_LC(BN_, OF_, 128); else NRPN_LC(BN_, OF_, 64); } while (0)
#define NRPN_LBIG(...)                                                                       \
  do {                                                                                       \
    NRPN_LDS((conv_igemm_big_kernel<__VA_ARGS__>), (int)lds_);                               \
    hipLaunchKernelGGL((conv_igemm_big_kernel<__VA_ARGS__>), grid, dim3(512), lds_, st, a);  \
  } while (0)
  if (kind == 5) {
    if constexpr (MODE == 0 && sizeof(T) == 2) {
      const size_t lds_ = 2 * (size_t)(256 + 256) * 128;
      if (out_f32) {
        NRPN_LDS((conv_igemm_big4_kernel<true>), (int)lds_);
        hipLaunchKernelGGL(conv_igemm_big4_kernel<true>, grid, dim3(256), lds_, st, a);
      } else {
        NRPN_LDS((conv_igemm_big4_kernel<false>), (int)lds_);
        hipLaunchKernelGGL(conv_igemm_big4_kernel<false>, grid, dim3(256), lds_, st, a);
      }
    }
  } else if (huge) {
    if constexpr (MODE == 0 && sizeof(T) == 2) {
      const size_t lds_ = 2 * (size_t)(256 + 256) * 128;
      const int dbg = ((kn.dbg & NRPN_CONV_DEBUG_ALIAS_TAPS) ? 1 : 0) | ((kn.dbg & NRPN_CONV_DEBUG_NO_SYNC) ? 2 : 0);
      if (!out_f32 && (kn.stagger || (kn.dbg & NRPN_CONV_DEBUG_STAGGER))) {
        // DBG != 0: tools-only timing variants (wrong results by construction); the production instantiation has no such branch
        if (dbg == 0) NRPN_LBIG(false, true, 0);
        else if (dbg == 1) NRPN_LBIG(false, true, 1);
        else if (dbg == 2) NRPN_LBIG(false, true, 2);
        else NRPN_LBIG(false, true, 3);
      } else if (out_f32) {
        NRPN_LBIG(true, false, 0);
      } else {
        NRPN_LBIG(false, false, 0);
      }
    }
  } else if (big) {
    if constexpr (MODE == 0 && sizeof(T) == 2) {
      const size_t lds_ = 3 * (size_t)(256 + 128) * 128;
      if (out_f32) {
        NRPN_LDS((conv_igemm_ws_kernel<true>), (int)lds_);
        hipLaunchKernelGGL(conv_igemm_ws_kernel<true>, grid, dim3(512), lds_, st, a);
      } else {
        NRPN_LDS((conv_igemm_ws_kernel<false>), (int)lds_);
        hipLaunchKernelGGL(conv_igemm_ws_kernel<false>, grid, dim3(512), lds_, st, a);
      }
    }
  } else if (bn == 64) { if (out_f32) NRPN_LC2(64, true); else NRPN_LC2(64, false); }
  else { if (out_f32) NRPN_LC2(128, true); else NRPN_LC2(128, false); }
#undef NRPN_LBIG
#undef NRPN_LC2
#undef NRPN_LC
  if (rc) return rc;
  NRPN_LAUNCH_CHECK("conv_igemm");
  return NRPN_OK;
}

static size_t fwd_workspace_bytes(long long M, int cin, int cout, int ksize, int dtype, const Knobs &kn, const Grid &g) {
  if (halo_ok(g, cin, cout, ksize == 3 ? 27 : 1, dtype == NRPN_F32 ? 4 : 2, false, kn)) return 0;
  const int bs = conv_big_split(M, cout, cin, ksize == 3 ? 27 : 1, dtype == NRPN_F32 ? 4 : 2, kn);
  if (bs) return (size_t)bs * M * cout * 4;
  const int s = conv_ksplit(M, cout, cin, ksize == 3 ? 27 : 1, dtype == NRPN_F32 ? 4 : 2, kn);
  if (s > 1) return (size_t)s * M * cout * 4;
  if (conv_tile_kind(M, cout, cin, dtype == NRPN_F32 ? 4 : 2, false, false, kn) == 1) {
    const TailSplit t = conv_tail_split(M, cout, cin, ksize == 3 ? 27 : 1, dtype == NRPN_F32 ? 4 : 2, kn);
    if (t.ks > 1) return (size_t)t.ks * (M - (long long)t.tile0 * 256) * cout * 4;
  }
  return 0;
}
extern "C" size_t nrpn_conv3d_fwd_workspace_bytes(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype) {
  return fwd_workspace_bytes((long long)n * gx * gy * gz, cin, cout, ksize, dtype, resolve_knobs(nullptr), Grid{n, gx, gy, gz});
}
extern "C" size_t nrpn_conv3d_fwd_workspace_bytes_ex(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype,
                                                     const nrpn_conv_opts *opts) {
  return fwd_workspace_bytes((long long)n * gx * gy * gz, cin, cout, ksize, dtype, resolve_knobs(opts), Grid{n, gx, gy, gz});
}

// Which kernel a forward / dgrad launch of this shape selects (mirrors launch_conv + conv3d_fwd_impl; lets tests assert that a
// shape really exercises the kernel they claim to cover): 0 = 128-row tile, 1 = 256x256 tile (8 waves), 2 = 256x256 tile on K slices,
// 3 = 128-row tile on K slices, 4 = wave-specialised 256x128 (opt-in), 5 = 256x256 tile on 4 waves, 6 = the same on K slices,
// 7 = halo form of the 3x3x3 kernel (4 x 8 x 8 voxel blocks).
static int fwd_plan(long long M, int cin, int cout, int ksize, int dtype, const Knobs &kn, const Grid &g) {
  const int es = dtype == NRPN_F32 ? 4 : 2, taps = ksize == 3 ? 27 : 1;
  if (halo_ok(g, cin, cout, taps, es, false, kn)) return 7;
  if (conv_big_split(M, cout, cin, taps, es, kn)) return conv_tile_kind(M, cout, cin, es, true, false, kn) == 5 ? 6 : 2;
  if (conv_ksplit(M, cout, cin, taps, es, kn) > 1) return 3;
  return conv_tile_kind(M, cout, cin, es, false, false, kn);
}
extern "C" int nrpn_conv3d_fwd_plan(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype) {
  return fwd_plan((long long)n * gx * gy * gz, cin, cout, ksize, dtype, resolve_knobs(nullptr), Grid{n, gx, gy, gz});
}
extern "C" int nrpn_conv3d_fwd_plan_ex(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype, const nrpn_conv_opts *opts) {
  return fwd_plan((long long)n * gx * gy * gz, cin, cout, ksize, dtype, resolve_knobs(opts), Grid{n, gx, gy, gz});
}

static int conv3d_fwd_impl(const void *x, const void *wp, const float *bias, const void *mask, void *y, long long M, int gx, int gy, int gz, const Segs *segs,
                           int cin, int cout, int wrows, int ksize, int dtype, int flags, void *workspace, nrpn_stream_t stream,
                           float *stats = nullptr, const nrpn_conv_opts *opts = nullptr) {
  NRPN_REQUIRE(ksize == 1 || ksize == 3, "conv3d_fwd: ksize must be 1 or 3 (got %d)", ksize);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "conv3d_fwd: bad dtype %d", dtype);
  NRPN_REQUIRE(M > 0 && gx > 0 && gy > 0 && gz > 0 && cin > 0 && cout > 0 && wrows >= cout, "conv3d_fwd: bad sizes");
  const int es = dtype == NRPN_F32 ? 4 : 2;
  NRPN_REQUIRE((cin * es) % 64 == 0, "conv3d_fwd: Cin*elemsize must be a multiple of 64 bytes (Cin=%d)", cin);
  NRPN_REQUIRE(x && wp && y, "conv3d_fwd: null pointer");
  const Knobs kn = resolve_knobs(opts, flags);
  ConvArgs a{};
  a.x = x; a.w = wp; a.bias = bias; a.mask = mask; a.y = y; a.stats = stats;
  a.scale = opts ? opts->scale : nullptr;
  a.M = M;
  a.X = gx; a.Y = gy; a.Z = gz; a.OX = gx; a.OY = gy; a.OZ = gz;   // classic layout: the kernel splits v into (batch, x, y, z) with these
  if (segs) a.segs = *segs;
  a.dvz = make_fastdiv((unsigned)gz); a.dvy = make_fastdiv((unsigned)gy);
  a.dvs = make_fastdiv((unsigned)min((long long)gx * gy * gz, (1ll << 31) - 1));      // a scene of >= 2^31 voxels does not pass the 2 GiB check below
  a.Cin = cin; a.Cout = cout; a.wrows = wrows; a.taps = ksize == 3 ? 27 : 1; a.stride = 1;
  a.flags = flags & 3;
  NRPN_REQUIRE(a.M * cin * es < (1ll << 31) && (long long)a.taps * wrows * cin * es < (1ll << 31),
               "conv3d_fwd: activation / weight tensors must stay below 2 GiB (32-bit buffer offsets)");
  a.x_bytes = (unsigned)(a.M * cin * es); a.w_bytes = (unsigned)((long long)a.taps * wrows * cin * es);
  const bool out_f32 = (flags & NRPN_CONV_OUT_F32) != 0;
  hipStream_t st = as_stream(stream);
  if (!segs && dtype == NRPN_BF16) {
    const Grid g{(int)(M / ((long long)gx * gy * gz)), gx, gy, gz};
    if (halo_ok(g, cin, cout, a.taps, es, out_f32, kn)) {
      const long long wgs = halo_tiles(g) * ((cout + 255) / 256);
      NRPN_REQUIRE(wgs < (1ll << 31), "conv3d_fwd: too many tiles");
      const bool xp = cin % 128 == 0 && !((kn.dbg >> 12) & 3) && (kn.halo_xp == 1 || (kn.halo_xp == 2 && cin >= 256));
      const int variant = ((kn.dbg >> 12) & 3) | (xp ? 4 : 0) | (out_f32 ? 8 : 0);
      return nrpn_launch_conv_halo(a, (unsigned)wgs, st, variant);
    }
  }
  const int bs = workspace ? conv_big_split(a.M, cout, cin, a.taps, es, kn) : 0;
  if (bs) {
    a.ksplit = bs; a.slices = 1; a.ws = reinterpret_cast<float *>(workspace);
  } else {
    a.ksplit = workspace ? conv_ksplit(a.M, cout, cin, a.taps, es, kn) : 1;
    if (a.ksplit > 1) a.ws = reinterpret_cast<float *>(workspace);
  }
  const int nsl = a.ksplit;
  if (a.ksplit <= 1 && workspace && !segs && !stats && !out_f32 && dtype == NRPN_BF16 && conv_tile_kind(a.M, cout, cin, es, false, false, kn) == 1) {
    const TailSplit t = conv_tail_split(a.M, cout, cin, a.taps, es, kn);
    if (t.ks > 1) { a.tail_tile0 = t.tile0; a.tail_ks = t.ks; a.ws = reinterpret_cast<float *>(workspace); }
  }
  int rc = (dtype == NRPN_F32) ? launch_conv<float, 0>(a, true, st, kn) : launch_conv<bf16s, 0>(a, out_f32, st, kn);
  if (!rc && a.tail_ks > 1) {      // bias / scale / ReLU / mask / cast of the tail rows, from their K-slice partials
    const long long row0 = (long long)a.tail_tile0 * 256, total = (a.M - row0) * cout;
    const int blocks = (int)min((long long)2048, (total + 255) / 256);
    const float *b = (flags & NRPN_CONV_BIAS) ? bias : nullptr;
    const bf16s *mk = a.mask ? reinterpret_cast<const bf16s *>(a.mask) + row0 * cout : nullptr;
    hipLaunchKernelGGL((splitk_epilogue_kernel<bf16s, false>), dim3(blocks), dim3(256), 0, st, a.ws, b, reinterpret_cast<bf16s *>(y) + row0 * cout, total, cout,
                       (flags & NRPN_CONV_RELU) ? 1 : 0, a.tail_ks, mk, a.scale);
    NRPN_LAUNCH_CHECK("splitk_epilogue (tail)");
    return NRPN_OK;
  }
  if (rc || a.ksplit <= 1) return rc;
  const long long total = a.M * cout;
  const int blocks = (int)min((long long)4096, (total + 255) / 256);
  const float *b = (flags & NRPN_CONV_BIAS) ? bias : nullptr;
  const int relu = (flags & NRPN_CONV_RELU) ? 1 : 0;
  if (dtype == NRPN_F32 || out_f32) {
    if (dtype == NRPN_F32) hipLaunchKernelGGL((splitk_epilogue_kernel<float, true>), dim3(blocks), dim3(256), 0, st, a.ws, b, y, total, cout, relu, nsl, (const float *)a.mask, a.scale);
    else hipLaunchKernelGGL((splitk_epilogue_kernel<bf16s, true>), dim3(blocks), dim3(256), 0, st, a.ws, b, y, total, cout, relu, nsl, (const bf16s *)a.mask, a.scale);
  } else {
    hipLaunchKernelGGL((splitk_epilogue_kernel<bf16s, false>), dim3(blocks), dim3(256), 0, st, a.ws, b, y, total, cout, relu, nsl, (const bf16s *)a.mask, a.scale);
  }
  NRPN_LAUNCH_CHECK("splitk_epilogue");
  return NRPN_OK;
}

extern "C" int nrpn_conv3d_fwd(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cin,
                               int cout, int wrows, int ksize, int dtype, int flags, void *workspace, const void *relu_mask,
                               nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0, "conv3d_fwd: bad sizes");
  NRPN_REQUIRE(!relu_mask || !(flags & NRPN_CONV_OUT_F32) || dtype == NRPN_F32, "conv3d_fwd: relu_mask needs outputs in the input dtype");
  return conv3d_fwd_impl(x, wp, bias, relu_mask, y, (long long)n * gx * gy * gz, gx, gy, gz, nullptr, cin, cout, wrows, ksize, dtype, flags,
                         workspace, stream);
}

// Rows P of the partial-statistics buffer [P][2][Cout] a forward launch of this shape fills when asked to (0 = the shape runs a kernel
// without fused statistics: K-sliced, fp32, narrow K-step or the opt-in variants -- use nrpn_bn_stats on the output instead).
static int fwd_stats_rows(long long M, int cin, int cout, int ksize, int dtype, const Knobs &kn, const Grid &g) {
  if (dtype != NRPN_BF16 || (cout & 7) != 0 || !kn.glds || kn.bm == 256) return 0;
  const int plan = fwd_plan(M, cin, cout, ksize, dtype, kn, g);
  if (plan == 7) return (int)(halo_tiles(g) * 2);
  if (plan == 1 || plan == 5) return (int)(cdiv64(M, 256) * 2);
  if (plan != 0 || kn.kb != 128 || (cin * 2) % 128 != 0) return 0;
  return (int)(cdiv64(M, 128) * (cout <= 64 ? 4 : 2));
}
extern "C" int nrpn_conv3d_fwd_stats_rows(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype) {
  return fwd_stats_rows((long long)n * gx * gy * gz, cin, cout, ksize, dtype, resolve_knobs(nullptr), Grid{n, gx, gy, gz});
}
extern "C" int nrpn_conv3d_fwd_stats_rows_ex(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype, const nrpn_conv_opts *opts) {
  return fwd_stats_rows((long long)n * gx * gy * gz, cin, cout, ksize, dtype, resolve_knobs(opts), Grid{n, gx, gy, gz});
}

// nrpn_conv3d_fwd + partial BatchNorm statistics of the stored outputs from the same launch (finish them with nrpn_bn_stats_finalize).
extern "C" int nrpn_conv3d_fwd_stats(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cin,
                                     int cout, int wrows, int ksize, int dtype, int flags, float *stats, nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0 && stats, "conv3d_fwd_stats: bad sizes / null statistics buffer");
  NRPN_REQUIRE(nrpn_conv3d_fwd_stats_rows(n, gx, gy, gz, cin, cout, ksize, dtype) > 0 && !(flags & NRPN_CONV_OUT_F32) && wrows == cout,
               "conv3d_fwd_stats: this shape does not run a kernel with fused statistics");
  return conv3d_fwd_impl(x, wp, bias, nullptr, y, (long long)n * gx * gy * gz, gx, gy, gz, nullptr, cin, cout, wrows, ksize, dtype, flags,
                         nullptr, stream, stats);
}

// The general form: nrpn_conv3d_fwd with a per-call plan and the fused-epilogue extras of `opts` (see nerfrpn.h): per-channel scale
// (eval-mode BatchNorm fold), ReLU mask, BatchNorm statistics partials, tile / K-step / staging overrides.  opts == NULL: nrpn_conv3d_fwd.
extern "C" int nrpn_conv3d_fwd_ex(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cin,
                                  int cout, int wrows, int ksize, int dtype, int flags, void *workspace, const nrpn_conv_opts *opts,
                                  nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0, "conv3d_fwd_ex: bad sizes");
  NRPN_REQUIRE(!opts || opts->size == (int32_t)sizeof(nrpn_conv_opts), "conv3d_fwd_ex: opts->size must be sizeof(nrpn_conv_opts)");
  const void *mask = opts ? opts->relu_mask : nullptr;
  float *stats = opts ? opts->stats : nullptr;
  NRPN_REQUIRE(!mask || !(flags & NRPN_CONV_OUT_F32) || dtype == NRPN_F32, "conv3d_fwd_ex: relu_mask needs outputs in the input dtype");
  if (stats) {
    NRPN_REQUIRE(fwd_stats_rows((long long)n * gx * gy * gz, cin, cout, ksize, dtype, resolve_knobs(opts), Grid{n, gx, gy, gz}) > 0 && !(flags & NRPN_CONV_OUT_F32) &&
                     wrows == cout && !mask,
                 "conv3d_fwd_ex: this shape / plan does not run a kernel with fused statistics (nrpn_conv3d_fwd_stats_rows_ex)");
    workspace = nullptr;
  }
  return conv3d_fwd_impl(x, wp, bias, mask, y, (long long)n * gx * gy * gz, gx, gy, gz, nullptr, cin, cout, wrows, ksize, dtype, flags,
                         workspace, stream, stats, opts);
}

extern "C" int nrpn_conv3d_fwd_ragged(const void *x, const void *wp, const float *bias, void *y, int nseg, const int32_t *dims, int cin,
                                      int cout, int wrows, int ksize, int dtype, int flags, void *workspace, nrpn_stream_t stream) {
  Segs sg{};
  long long M = 0;
  if (int rc = fill_segs(sg, nseg, dims, M)) return rc;
  return conv3d_fwd_impl(x, wp, bias, nullptr, y, M, 1, 1, 1, &sg, cin, cout, wrows, ksize, dtype, flags, workspace, stream);
}

// Row-list form of the forward / dgrad launch: output rows = the `nrows` voxels of `rows` ([nrows][2] u32 = {voxel id in the ragged space of
// `dims`, tap word}; csrc/cone.hip builds such lists) -- x, y and relu_mask are indexed by voxel id, rows outside the list are neither
// read as outputs nor written.  128-row tiles of conv_igemm_kernel, no K slices (the lists are short: one launch, no workspace).
// tools-only A/B switch (include/nerfrpn_tools.h): 256x256 tile for long row lists.  OFF by default -- measured on one box, bench step,
// S3 = 33 k rows = 130 tiles: 9.47 ms with it against 9.43 ms on 128-row tiles (three alternating runs each): the 8-wave tile's
// chunk-outer K order counts on an XCD's workgroups sweeping one contiguous voxel slab through its L2, which a gathered list is not.
static std::atomic<int> g_rows_big{0};
extern "C" int nrpn_set_rows_big_tile(int on) { g_rows_big = on ? 1 : 0; return NRPN_OK; }

// Short lists (the S0 / S1 cones: 2-70 tiles) would leave the chip idle behind a 54-step K loop: they run on K slices whose fp32 partials
// ([slices][nrows][Cout], plain stores) are summed in slice order and scattered by rows_epilogue_kernel.
static int rows_ksplit(long long nrows, int cin, int cout, int taps, int es) {
  const long long tiles = cdiv64(nrows, 128) * ((cout + 127) / 128);
  const int nk = taps * (cin * es / 128);
  if (tiles >= 128 || nk < 12) return 1;
  long long s = (256 + tiles - 1) / tiles;
  if (s > nk / 6) s = nk / 6;
  if (s > 32) s = 32;
  if (s < 2) return 1;
  const long long per = (nk + s - 1) / s;
  s = (nk + per - 1) / per;                  // no empty slice
  return s < 2 ? 1 : (int)s;
}
// Long lists: the 128-row tiling of the S3 cone is a little more than one round of the chip's 512 workgroup slots (33-35 k rows x 2 column
// tiles = 518-552 workgroups), i.e. two rounds in time.  The M tiles of the short last round run on K slices (cf. conv_tail_split).
static std::atomic<int> g_rows_tail{1};     // tools-only A/B switch
extern "C" int nrpn_set_rows_tail_split(int on) { g_rows_tail = on ? 1 : 0; return NRPN_OK; }
static TailSplit rows_tail_split(long long nrows, int cin, int cout, int taps, int es) {
  TailSplit none{0, 0};
  if (!g_rows_tail.load(std::memory_order_relaxed)) return none;
  const long long mt = cdiv64(nrows, 128), nt = (cout + 127) / 128, tiles = mt * nt;
  const long long rem = tiles % 512;
  if (tiles <= 512 || rem == 0 || rem > 256) return none;
  const long long tail_mt = (rem + nt - 1) / nt;
  const int nk = taps * (cin * es / 128);
  int ks = (int)(512 / (tail_mt * nt));
  if (ks > 8) ks = 8;
  while (ks > 1 && nk / ks < 6) --ks;
  if (ks < 2) return none;
  const int per = (nk + ks - 1) / ks;
  ks = (nk + per - 1) / per;
  return ks >= 2 ? TailSplit{(int)(mt - tail_mt), ks} : none;
}
extern "C" size_t nrpn_conv3d_fwd_rows_workspace_bytes(int64_t nrows, int cin, int cout, int ksize, int dtype) {
  const int es = dtype == NRPN_F32 ? 4 : 2, taps = ksize == 3 ? 27 : 1;
  const int s = rows_ksplit(nrows, cin, cout, taps, es);
  if (s > 1) return (size_t)s * nrows * cout * 4;
  const TailSplit t = rows_tail_split(nrows, cin, cout, taps, es);
  return t.ks > 1 ? (size_t)t.ks * (nrows - (long long)t.tile0 * 128) * cout * 4 : 0;
}

extern "C" int nrpn_conv3d_fwd_rows(const void *x, const void *wp, const float *bias, void *y, const uint32_t *rows, int64_t nrows, int nseg,
                                    const int32_t *dims, int cin, int cout, int wrows, int ksize, int dtype, int flags, const void *relu_mask,
                                    void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(ksize == 1 || ksize == 3, "conv3d_fwd_rows: ksize must be 1 or 3 (got %d)", ksize);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "conv3d_fwd_rows: bad dtype %d", dtype);
  NRPN_REQUIRE(x && wp && y && rows && nrows >= 0 && cin > 0 && cout > 0 && wrows >= cout, "conv3d_fwd_rows: bad arguments");
  if (nrows == 0) return NRPN_OK;
  const int es = dtype == NRPN_F32 ? 4 : 2;
  NRPN_REQUIRE((cin * es) % 128 == 0, "conv3d_fwd_rows: Cin*elemsize must be a multiple of 128 bytes (Cin=%d)", cin);
  const bool out_f32 = (flags & NRPN_CONV_OUT_F32) != 0 || dtype == NRPN_F32;
  NRPN_REQUIRE(!relu_mask || !(flags & NRPN_CONV_OUT_F32) || dtype == NRPN_F32, "conv3d_fwd_rows: relu_mask needs outputs in the input dtype");
  ConvArgs a{};
  long long total = 0;
  if (int rc = fill_segs(a.segs, nseg, dims, total)) return rc;
  a.x = x; a.w = wp; a.bias = bias; a.mask = relu_mask; a.y = y; a.rows = rows;
  a.M = nrows;
  a.X = a.Y = a.Z = a.OX = a.OY = a.OZ = 1;
  a.Cin = cin; a.Cout = cout; a.wrows = wrows; a.taps = ksize == 3 ? 27 : 1; a.stride = 1; a.flags = flags & 3; a.ksplit = 1;
  NRPN_REQUIRE(total * cin * es < (1ll << 31) && total * cout * 4 < (1ll << 32) && (long long)a.taps * wrows * cin * es < (1ll << 31),
               "conv3d_fwd_rows: activation / weight tensors must stay below 2 GiB (32-bit buffer offsets)");
  a.x_bytes = (unsigned)(total * cin * es); a.w_bytes = (unsigned)((long long)a.taps * wrows * cin * es);
  hipStream_t st = as_stream(stream);
  const int ks = workspace ? rows_ksplit(nrows, cin, cout, a.taps, es) : 1;
  if (ks > 1) { a.ksplit = ks; a.ws = reinterpret_cast<float *>(workspace); }
  // opt-in (nrpn_set_rows_big_tile): long bf16 lists (>= 96 tiles of 256 rows) of wide layers on the 256x256 tile
  if (g_rows_big.load(std::memory_order_relaxed) && dtype == NRPN_BF16 && !out_f32 && ks == 1 && cout >= 256 && (cout & 7) == 0 &&
      cdiv64(nrows, 256) * ((cout + 255) / 256) >= 96) {
    const size_t lds_big = 2 * (size_t)(256 + 256) * 128;
    dim3 gbig((unsigned)(cdiv64(nrows, 256) * ((cout + 255) / 256)));
    NRPN_LDS((conv_igemm_big_kernel<false, true, 0, true>), (int)lds_big);
    hipLaunchKernelGGL((conv_igemm_big_kernel<false, true, 0, true>), gbig, dim3(512), lds_big, st, a);
    NRPN_LAUNCH_CHECK("conv3d_fwd_rows (256x256 tile)");
    return NRPN_OK;
  }
  dim3 grid((unsigned)(cdiv64(nrows, 128) * ((cout + 127) / 128) * ks));
  if (ks == 1 && workspace) {
    const TailSplit t = rows_tail_split(nrows, cin, cout, a.taps, es);
    if (t.ks > 1) {
      a.tail_tile0 = t.tile0; a.tail_ks = t.ks; a.ws = reinterpret_cast<float *>(workspace);
      const long long mt = cdiv64(nrows, 128), nt = (cout + 127) / 128;
      grid = dim3((unsigned)(t.tile0 * nt + (mt - t.tile0) * nt * t.ks));
    }
  }
  const size_t lds_ = 2 * (size_t)(128 + 128) * 128;
  int rc;
  if (dtype == NRPN_F32) rc = launch_igemm(conv_igemm_kernel<float, 128, 0, true, 128, true, 128, true>, grid, lds_, st, a);
  else if (out_f32) rc = launch_igemm(conv_igemm_kernel<bf16s, 128, 0, true, 128, true, 128, true>, grid, lds_, st, a);
  else rc = launch_igemm(conv_igemm_kernel<bf16s, 128, 0, false, 128, true, 128, true>, grid, lds_, st, a);
  if (rc) return rc;
  NRPN_LAUNCH_CHECK("conv3d_fwd_rows");
  if (ks > 1 || a.tail_ks > 1) {
    const long long row0 = a.tail_ks > 1 ? (long long)a.tail_tile0 * 128 : 0;      // the partials cover the rows from row0 on
    const int nsl = a.tail_ks > 1 ? a.tail_ks : ks;
    const long long total = (nrows - row0) * cout;
    const int blocks = (int)min((long long)2048, (total + 255) / 256);
    const float *b = (flags & NRPN_CONV_BIAS) ? bias : nullptr;
    const int relu = (flags & NRPN_CONV_RELU) ? 1 : 0;
    const uint32_t *rw = rows + 2 * row0;
    if (dtype == NRPN_F32) hipLaunchKernelGGL((rows_epilogue_kernel<float, true>), dim3(blocks), dim3(256), 0, st, a.ws, b, y, total, cout, relu, nsl, (const float *)relu_mask, rw);
    else if (out_f32) hipLaunchKernelGGL((rows_epilogue_kernel<bf16s, true>), dim3(blocks), dim3(256), 0, st, a.ws, b, y, total, cout, relu, nsl, (const bf16s *)relu_mask, rw);
    else hipLaunchKernelGGL((rows_epilogue_kernel<bf16s, false>), dim3(blocks), dim3(256), 0, st, a.ws, b, y, total, cout, relu, nsl, (const bf16s *)relu_mask, rw);
    NRPN_LAUNCH_CHECK("rows_epilogue");
  }
  return NRPN_OK;
}

static int stem_fwd_impl(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cout,
                         int stride, int dtype, int flags, const nrpn_conv_opts *opts, nrpn_stream_t stream) {
  NRPN_REQUIRE(stride == 1 || stride == 2, "stem: stride must be 1 or 2 (got %d)", stride);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "stem: bad dtype %d", dtype);
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0 && cout > 0, "stem: bad sizes");
  NRPN_REQUIRE(x && wp && y, "stem: null pointer");
  NRPN_REQUIRE(!opts || opts->size == (int32_t)sizeof(nrpn_conv_opts), "stem: opts->size must be sizeof(nrpn_conv_opts)");
  NRPN_REQUIRE(!opts || (!opts->relu_mask && !opts->stats), "stem: relu_mask / stats are not available on the stem");
  ConvArgs a{};
  a.x = x; a.w = wp; a.bias = bias; a.y = y;
  a.scale = opts ? opts->scale : nullptr;
  a.X = gx; a.Y = gy; a.Z = gz;
  a.OX = (gx + 6 - 7) / stride + 1; a.OY = (gy + 6 - 7) / stride + 1; a.OZ = (gz + 6 - 7) / stride + 1;
  a.M = (long long)n * a.OX * a.OY * a.OZ;
  a.Cin = 4; a.Cout = cout; a.wrows = cout; a.taps = 343; a.stride = stride; a.flags = flags & 3;
  {
    const long long xb = (long long)n * gx * gy * gz * 4 * (dtype == NRPN_F32 ? 4 : 2);
    NRPN_REQUIRE(xb < (1ll << 31), "stem: input must stay below 2 GiB");
    a.x_bytes = (unsigned)xb; a.w_bytes = 0;
  }
  const Knobs kn = resolve_knobs(opts);
  if (dtype == NRPN_F32) return launch_conv<float, 1>(a, true, as_stream(stream), kn);
  return launch_conv<bf16s, 1>(a, false, as_stream(stream), kn);
}

extern "C" int nrpn_conv3d_stem_fwd(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cout,
                                    int stride, int dtype, int flags, nrpn_stream_t stream) {
  return stem_fwd_impl(x, wp, bias, y, n, gx, gy, gz, cout, stride, dtype, flags, nullptr, stream);
}

// nrpn_conv3d_stem_fwd with the fused-epilogue extras of `opts` that apply to the stem: per-channel scale (eval-mode BatchNorm fold)
// with NRPN_CONV_RELU on top (feature_extractor.py:336-338 conv -> BN -> ReLU in one launch)
extern "C" int nrpn_conv3d_stem_fwd_ex(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cout,
                                       int stride, int dtype, int flags, const nrpn_conv_opts *opts, nrpn_stream_t stream) {
  return stem_fwd_impl(x, wp, bias, y, n, gx, gy, gz, cout, stride, dtype, flags, opts, stream);
}


// =====================================================================================================================
// Stem forward, "halo" form: Conv3d(4 -> 64, k7, stride 2, pad 3) on bf16 with even Z  (feature_extractor.py:336, VGG / ResNet stems)
//
// The im2col kernel above pulls 343 taps x 8 bytes = 2.7 kB through the L1 -> LDS path per output voxel although neighbouring outputs
// share almost all of it (1.4 GB per 160^3 scene, 200 us).  Here a workgroup owns a 4 x 4 x 16 block of output voxels, stages the
// 13 x 13 x 40 x 4-channel input halo of that block in LDS ONCE (57 KB: ~230 B per output voxel) and takes every A fragment straight
// from it: for a fixed (dx, dy) the 7 dz taps x 4 channels of one output voxel are 28 contiguous elements of a z-row, 32 with one
// unused leading position (stride 2 => the run of output oz starts at the even z = 2 oz - 4), so lane (row, k-half) of the 32x32x16
// MFMA reads its 8 k-values as ONE ds_read_b128 at  halo[2 oxl + dx][2 oyl + dy][2 ozl + 4 s + 2 k-half]  -- a per-lane base plus a
// compile-time offset (the 25 K-steps are unrolled).  K = 50 (dx, dy) slots x 32 = 1600 (slot 49 and the leading z position carry zero
// weights: 86 % of the MFMA work is useful).  B = the weights [64][1600], streamed 2 taps (8 KB) per K-step by LDS-DMA, double-buffered.
// Row order inside the tile: r = ozl + 16 (oxl & 1) + 32 (oyl + 4 (oxl >> 1)), so the two half-blocks of a 32-row MFMA block are two
// x-rows = 35 x 256 bytes apart: conflict-free ds_read_b128.  74.8 KB of LDS: two workgroups per CU.
// =====================================================================================================================
struct StemHaloArgs {
  const void *x, *w;
  const float *bias, *scale;
  void *y;
  int N, X, Y, Z, OX, OY, OZ;
  int tx, ty, tz;           // tiles per axis
  int flags;
  unsigned x_bytes, w_bytes;
};
namespace sh {
constexpr int TX = 4, TY = 4, TZ = 16, HX = 13, HY = 14, PITCH = 320, PPRZ = PITCH / 16;     // HY: 13 used rows + 1 pad (x-row stride = 35 * 128 B)
constexpr int HALO_PIECES = HX * HY * PPRZ;                    // 3640 16-byte pieces
constexpr int HALO_INSTR = (HALO_PIECES + 63) / 64;            // 57 wave-instructions of 1 KiB
constexpr int HALO_BYTES = HALO_INSTR * 1024;                  // 58368
constexpr int KTAPS = 50, KROW = KTAPS * 32;                   // packed weight row length (elements)
constexpr int B_BYTES = 64 * 128;                              // one K-step: 64 output rows x 2 taps x 64 B
constexpr int LDS_BYTES = HALO_BYTES + 2 * B_BYTES;            // 74752
}  // namespace sh

__global__ void __launch_bounds__(256, 2) stem_fwd_halo_kernel(const StemHaloArgs p) {
  using namespace sh;
  typedef bf16s T;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char *halo = lds;
  char *Bt = lds + HALO_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // tile -> (n, ox0, oy0, oz0); consecutive workgroups walk z, then y, then x of one scene
  unsigned t = blockIdx.x;
  const int tzi = (int)(t % p.tz); t /= p.tz;
  const int tyi = (int)(t % p.ty); t /= p.ty;
  const int txi = (int)(t % p.tx);
  const int n = (int)(t / p.tx);
  const int ox0 = txi * TX, oy0 = tyi * TY, oz0 = tzi * TZ;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), wr = make_rsrc(p.w, p.w_bytes);
  // ---- halo: piece q = 64 * instr + lane -> (z-row, 2-z piece); whole pieces are inside or outside the grid (Z even)
#pragma unroll 1
  for (int inst = wave_u; inst < HALO_INSTR; inst += 4) {
    const int q = inst * 64 + lane;
    const int row = q / PPRZ, pz = q - row * PPRZ;
    const int hx = row / HY, hy = row - hx * HY;
    const int ix = 2 * ox0 - 3 + hx, iy = 2 * oy0 - 3 + hy, iz = 2 * oz0 - 4 + 2 * pz;
    const bool ok = q < HALO_PIECES && hy < 13 && (unsigned)ix < (unsigned)p.X && (unsigned)iy < (unsigned)p.Y && iz >= 0 && iz + 2 <= p.Z;
    const unsigned off = ok ? (unsigned)(((((long long)n * p.X + ix) * p.Y + iy) * p.Z + iz) * 8) : kOOB;
    lds_dma16(xr, halo + inst * 1024, off);
  }
  // ---- B tile of K-step ks: 64 rows x 8 pieces; this lane owns rows tid / 8 and tid / 8 + 32, physical slot tid % 8
  const int br = tid >> 3, bs = tid & 7;
  unsigned b_voff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = br + 32 * i;
    b_voff[i] = (unsigned)(r * KROW * 2) + ((bs ^ ((r >> 1) & 7)) << 4);
  }
  auto issue_b = [&](int buf, int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_dma16(wr, Bt + buf * B_BYTES + (wave_u * 8 + 32 * i) * 128, b_voff[i] + (unsigned)(ks * 128));
  };
  issue_b(0, 0);

  // ---- fragment addresses
  const int fr = lane & 31, fk = lane >> 5;
  int a_base[2], b_row[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int blk = wave * 2 + i;
    const int ozl = fr & 15, xh = fr >> 4, oyl = blk & 3, oxl = 2 * (blk >> 2) + xh;
    a_base[i] = ((2 * oxl) * HY + 2 * oyl) * PITCH + 16 * ozl + 16 * fk;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) b_row[j] = j * 32 + fr;
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < KTAPS / 2; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < KTAPS / 2) issue_b(buf ^ 1, ks + 1);
    const char *B = Bt + buf * B_BYTES;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int tap = 2 * ks + tt;
      if (tap >= 49) continue;              // slot 49: zero weights, and its halo offset would leave the staged block
      const int dx = tap / 7, dy = tap % 7;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        f4 af[2], bfv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const f4 *>(halo + a_base[i] + (dx * HY + dy) * PITCH + 32 * s2);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          bfv[j] = *reinterpret_cast<const f4 *>(B + b_row[j] * 128 + ((((tt * 2 + s2) * 2 + fk) ^ ((b_row[j] >> 1) & 7)) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], af[i], bfv[j]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: scale / bias / ReLU, staged through LDS (the halo is free after the last barrier), 16-byte stores
  const bool has_bias = (p.flags & NRPN_CONV_BIAS) && p.bias;
  const bool relu = p.flags & NRPN_CONV_RELU;
  constexpr int SP = 144;
  char *stage = lds + wave * (64 * SP);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = j * 32 + fr;
    const float bv = has_bias ? p.bias[col] : 0.f;
    const float sv = p.scale ? p.scale[col] : 1.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float o = acc[i][j][r] * sv + bv;
        if (relu) o = fmaxf(o, 0.f);
        *reinterpret_cast<bf16s *>(stage + (i * 32 + frag_row(r, lane)) * SP + col * 2) = f32_to_bf16_bits(o);
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  T *yp = reinterpret_cast<T *>(p.y);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int pc = lane + 64 * q;                 // 64 rows x 8 pieces
    const int rl = pc >> 3, seg = pc & 7;
    const int rr = rl & 31, blk = wave * 2 + (rl >> 5);
    const int oz = oz0 + (rr & 15), oy = oy0 + (blk & 3), ox = ox0 + 2 * (blk >> 2) + (rr >> 4);
    if (ox < p.OX && oy < p.OY && oz < p.OZ) {
      const f4 val = *reinterpret_cast<const f4 *>(stage + rl * SP + seg * 16);
      *reinterpret_cast<f4 *>(yp + ((((long long)n * p.OX + ox) * p.OY + oy) * p.OZ + oz) * 64 + seg * 8) = val;
    }
  }
}

// halo-form stem weights: reference [64][4][7][7][7] fp32 -> bf16 [64][50 * 32], k = (dx * 7 + dy) * 32 + (dz + 1) * 4 + c; the unused
// leading z position of every slot and slot 49 are zero
__global__ void pack_stem_halo_kernel(const float *__restrict__ w, int cout, bf16s *__restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)cout * sh::KROW) return;
  const int k = (int)(i % sh::KROW), o = (int)(i / sh::KROW);
  const int slot = k >> 5, zc = k & 31, zrel = zc >> 2, c = zc & 3;
  float v = 0.f;
  if (slot < 49 && zrel >= 1) v = w[((long long)o * 4 + c) * 343 + slot * 7 + (zrel - 1)];
  out[i] = f32_to_bf16_bits(v);
}

extern "C" int nrpn_stem_halo_supported(int gz, int cout, int stride, int dtype) {
  return (dtype == NRPN_BF16 && stride == 2 && cout == 64 && gz % 2 == 0) ? 1 : 0;
}
extern "C" int nrpn_stem_halo_kpad(void) { return sh::KROW; }

extern "C" int nrpn_pack_stem_weight_halo(const float *w_ref, int cout, void *wp, nrpn_stream_t stream) {
  NRPN_REQUIRE(w_ref && wp && cout == 64, "pack_stem_weight_halo: Cout must be 64");
  hipLaunchKernelGGL(pack_stem_halo_kernel, dim3((unsigned)cdiv64((long long)cout * sh::KROW, 256)), dim3(256), 0, as_stream(stream), w_ref, cout,
                     (bf16s *)wp);
  NRPN_LAUNCH_CHECK("pack_stem_weight_halo");
  return NRPN_OK;
}

// y [N, (X+1)/2, (Y+1)/2, Z/2, 64] bf16 = stem conv of x [N,X,Y,Z,4] bf16 with wp from nrpn_pack_stem_weight_halo; epilogue:
// acc * scale + bias, ReLU (NRPN_CONV_BIAS / NRPN_CONV_RELU in flags; scale optional = eval-mode BatchNorm fold)
extern "C" int nrpn_conv3d_stem_fwd_halo(const void *x, const void *wp, const float *bias, const float *scale, void *y, int n, int gx, int gy,
                                         int gz, int cout, int flags, nrpn_stream_t stream) {
  NRPN_REQUIRE(nrpn_stem_halo_supported(gz, cout, 2, NRPN_BF16), "stem_fwd_halo: needs bf16, stride 2, Cout 64 and an even Z (got Z=%d Cout=%d)", gz, cout);
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0 && x && wp && y, "stem_fwd_halo: bad sizes / null pointer");
  StemHaloArgs a{};
  a.x = x; a.w = wp; a.bias = bias; a.scale = scale; a.y = y;
  a.N = n; a.X = gx; a.Y = gy; a.Z = gz;
  a.OX = (gx - 1) / 2 + 1; a.OY = (gy - 1) / 2 + 1; a.OZ = (gz - 1) / 2 + 1;
  a.tx = (a.OX + sh::TX - 1) / sh::TX; a.ty = (a.OY + sh::TY - 1) / sh::TY; a.tz = (a.OZ + sh::TZ - 1) / sh::TZ;
  a.flags = flags & 3;
  const long long xb = (long long)n * gx * gy * gz * 8;
  NRPN_REQUIRE(xb < (1ll << 31), "stem_fwd_halo: input must stay below 2 GiB");
  a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)(cout * sh::KROW * 2);
  const long long tiles = (long long)n * a.tx * a.ty * a.tz;
  NRPN_REQUIRE(tiles < (1ll << 31), "stem_fwd_halo: too many tiles");
  NRPN_LDS(stem_fwd_halo_kernel, sh::LDS_BYTES);
  hipLaunchKernelGGL(stem_fwd_halo_kernel, dim3((unsigned)tiles), dim3(256), sh::LDS_BYTES, as_stream(stream), a);
  NRPN_LAUNCH_CHECK("stem_fwd_halo");
  return NRPN_OK;
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
  float *gw;          // partial gradients, one per voxel slice: MODE 0 [ksplit][taps][wrows][Cin]; MODE 1 [ksplit][wrows][Kpad]
  long long M;        // voxels of dY (N*OX*OY*OZ)
  int X, Y, Z, OX, OY, OZ;
  int Cin, Cout, wrows, taps, stride;
  int ksplit;         // voxel range is cut into ksplit slices (blockIdx.z / taps)
  int kpad;           // MODE 1
  int ntiles_n;       // cin (MODE 0) / k (MODE 1) tiles
  const unsigned *vmask;   // MODE 0, taps == 27: per-voxel 27-bit mask of in-bounds taps (built by tap_mask_kernel)
  unsigned x_bytes, dy_bytes;
  float *gbias;       // optional: per-slice column sums of dY [ksplit][wrows], written (plain stores) by the (centre tap, first n-tile)
                      // workgroups from their LDS A tiles; bias_finalize_kernel sums the slices in order (deterministic)
  long long slice_stride;   // elements between the partial gradients of consecutive voxel slices ([ksplit][...] layout of gw)
  Segs segs;          // MODE 0: ragged voxel list (n > 0); the tap-mask word then carries the segment id in bits 27-31
  const unsigned *rows;   // ROWS kernels: K row k is voxel rows[2 k] of the ragged space with tap word rows[2 k + 1] (csrc/cone.hip); M = list
                          // length; x / dy are indexed by voxel id.  Bit 13 of the word (centre tap) marks a valid row.
};

template <typename T> struct WgCfg;
template <> struct WgCfg<float> { static constexpr int KV = 32, RS = 128 * 4; };
template <> struct WgCfg<bf16s> { static constexpr int KV = 64, RS = 128 * 2; };

// Unpadded [voxel][channel] tile with the 16-byte slot index XOR-ed by (row & 3) << 2: LDS-DMA needs a linear destination
// (1 KiB = 2-4 whole rows per wave-instruction), and the XOR spreads the four voxel rows a transpose-read lane group touches
// over all 16 slots of a 256-byte bank row (conflict-free for ds_read_b64_tr_b16, ds_read_b32 and the 16-byte stores).
__device__ __forceinline__ int wg_off(int row, int byte_in_row, int rs) {
  return row * rs + ((((byte_in_row >> 4) ^ ((row & 3) << 2))) << 4) + (byte_in_row & 15);
}

// one 32(channel) x 16-byte K fragment out of a [voxel][channel] LDS tile
template <typename T, bool TR>
__device__ __forceinline__ f4 wg_frag(const char *tile, int ctile0, int kbase, int lane);

template <typename T, bool TR>
__device__ __forceinline__ f4 wg_frag(const char *tile, int ctile0, int kbase, int lane) {
  if (sizeof(T) == 4) {
    // 16 bytes = 4 k-values for lane-half h: voxels kbase + 4h .. +3 (the same permutation on A and B)
    constexpr int RS = WgCfg<float>::RS;
    const int c = ctile0 + (lane & 31), h = lane >> 5;
    f4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float *>(tile + wg_off(kbase + 4 * h + q, c * 4, RS));
    return v;
  } else {
    // 16 bytes = 8 bf16 k-values for lane-half h: voxels kbase + 8h .. +7, channel ctile0 + (lane & 31)
    constexpr int RS = WgCfg<bf16s>::RS;
    const int h = lane >> 5;
    if (TR) {
      // transpose read: inside a 16-lane group, lane i receives element (i & 3) of the 8-byte chunks addressed by lanes
      // 4j + (i >> 2), j = 0..3.  Lane p therefore addresses voxel (p >> 2), channels cbase + 4 (p & 3) .. +3.
      const int p = lane & 15;
      const int cbase = ctile0 + 16 * ((lane >> 4) & 1);
      const int row = kbase + 8 * h + (p >> 2);            // row + 4 has the same (row & 3): one offset serves both reads
      const char *a0 = tile + wg_off(row, (cbase + 4 * (p & 3)) * 2, RS);
      const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0));
      const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0 + 4 * RS));
      typedef __attribute__((ext_vector_type(2))) long long l2v;
      l2v r = {__builtin_bit_cast(long long, lo), __builtin_bit_cast(long long, hi)};
      return __builtin_bit_cast(f4, r);
    }
    const int c = ctile0 + (lane & 31);
    typedef __attribute__((ext_vector_type(8))) unsigned short u8v;
    u8v r;
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = *reinterpret_cast<const unsigned short *>(tile + wg_off(kbase + 8 * h + q, c * 2, RS));
    return __builtin_bit_cast(f4, r);
  }
}

// PACK2 (round 5, MODE 0 dense k3 only): layers with Cin <= 64 fill a quarter (Cout 64) or half (Cout 128) of the 128 x 128 tile and ran at
// 150 / 290 TFLOP/s.  A workgroup then owns a PAIR of taps: the B tile's 128 columns hold [tap 2t: <= 64 channels | tap 2t + 1: <= 64 channels]
// -- one shared dY tile, two shifted X rows per voxel row -- so the wave column wn IS the tap of the pair; 14 workgroups per (tile, slice)
// instead of 27 (tap 27 does not exist: its half reads zeros and is not stored).
template <typename T, int MODE, bool TR, bool ROWS = false, bool PACK2 = false>
__global__ void __launch_bounds__(256) conv_wgrad_kernel(const WgradArgs p) {
  static_assert(!PACK2 || (MODE == 0 && !ROWS), "PACK2 is a dense MODE 0 form");
  constexpr int KV = WgCfg<T>::KV, RS = WgCfg<T>::RS;
  constexpr int TILE = KV * RS;
  constexpr int PIECES_ROW = 128 * (int)sizeof(T) / 16;       // 16-byte pieces per 128-channel row: 16 (bf16) / 32 (fp32)
  constexpr int PIECES = KV * PIECES_ROW / 256;                // per thread per tile: 2 / 4
  constexpr int KSUB = (sizeof(T) == 2) ? 16 : 8;              // voxels consumed per fragment pair
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 buffers][A tile | B tile]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // 1-D grid, voxel slice slowest: id = ((slice * taps + tap) * ntn + ntile) * ntm + mtile.  All workgroups of a slice read the
  // same dY / X rows, and the XCD remap keeps a slice on one XCD's L2.
  const unsigned ntm = (p.wrows + 127) / 128, ntn = p.ntiles_n, tps = PACK2 ? 14u : ((MODE == 0) ? (unsigned)p.taps : 1u);
  unsigned id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (int)(id % ntm) * 128;   // cout tile
  id /= ntm;
  const int n0 = (int)(id % ntn) * 128;   // cin tile (MODE 0) / k tile (MODE 1); PACK2: 0
  id /= ntn;
  const int tap = PACK2 ? 2 * (int)(id % tps) : (int)(id % tps);      // PACK2: the pair's first tap; the second is tap + 1 (< 27 or absent)
  const int slice = (int)(id / tps);
  int dx = 0, dy = 0, dz = 0;
  if (MODE == 0 && p.taps == 27) { dx = tap / 9 - 1; dy = (tap / 3) % 3 - 1; dz = tap % 3 - 1; }
  const bool tap1_ok = PACK2 && tap + 1 < 27;

  const long long chunks = (p.M + KV - 1) / KV;
  const long long per = (chunks + p.ksplit - 1) / p.ksplit;
  const long long c_begin = slice * per, c_end = min(chunks, c_begin + per);
  if (c_begin >= c_end) return;

  const T *xbase = reinterpret_cast<const T *>(p.x);
  const T *dybase = reinterpret_cast<const T *>(p.dy);

  f4 ra[PIECES], rb[PIECES];

  // MODE 0 loader: raw-buffer loads (out-of-range offsets read zeros, so tail rows / border taps / column tails need no
  // branches); border validity of a voxel for this workgroup's tap comes from a per-voxel 27-bit mask, fetched one chunk
  // ahead so the mask -> address dependency never stalls the data loads.
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), dyr = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t mr = ROWS ? make_rsrc(p.rows, (unsigned)(p.M * 8)) : make_rsrc(p.vmask, (unsigned)(p.M * 4));
  // byte shift of this workgroup's tap inside each segment's own grid (classic layout: one entry); the mask word of a voxel
  // carries its segment id in bits 27..31
  __shared__ int seg_shift[kMaxSeg];
  __shared__ int seg_shift1[PACK2 ? kMaxSeg : 1];       // PACK2: the pair's second tap
  if (MODE == 0 && tid < kMaxSeg) {
    int Y = p.Y, Z = p.Z;
#pragma unroll
    for (int q = 0; q < kMaxSeg; ++q)
      if (q == tid && q < p.segs.n) { Y = p.segs.Y[q]; Z = p.segs.Z[q]; }
    seg_shift[tid] = (int)((((long long)dx * Y + dy) * Z + dz) * p.Cin * (long long)sizeof(T));
    if (PACK2) {
      const int t1 = tap + 1, dx1 = t1 / 9 - 1, dy1 = (t1 / 3) % 3 - 1, dz1 = t1 % 3 - 1;
      seg_shift1[tid] = (int)((((long long)dx1 * Y + dy1) * Z + dz1) * p.Cin * (long long)sizeof(T));
    }
  }
  __syncthreads();
  unsigned a_voff[PIECES], b_voff[PIECES], m_voff[PIECES], m_next[PIECES];
  bool b_half[PIECES];           // PACK2: this thread's B piece i belongs to the pair's second tap
#pragma unroll
  for (int i = 0; i < PIECES; ++i) b_half[i] = false;
  unsigned r_vox[PIECES];        // ROWS: voxel id of the row this piece belongs to in the NEXT chunk (fetched with its tap word)
  typedef __attribute__((ext_vector_type(2))) unsigned int rowpair;
  const bool use_mask = MODE == 0 && p.taps == 27;
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int pc = tid + 256 * i;
    const int row = pc / PIECES_ROW, col = (pc % PIECES_ROW) ^ ((row & 3) << 2);   // logical column of physical slot pc % PIECES_ROW
    const long long v = c_begin * KV + row;
    const int ca = m0 + col * (16 / (int)sizeof(T));
    // PACK2: logical pieces [0, PIECES_ROW / 2) of a B row belong to the pair's first tap, the rest to the second; both start at channel 0
    const int cb = PACK2 ? (col % (PIECES_ROW / 2)) * (16 / (int)sizeof(T)) : n0 + col * (16 / (int)sizeof(T));
    if (PACK2) b_half[i] = col >= PIECES_ROW / 2;
    r_vox[i] = 0;
    if (MODE == 0 && ROWS) {     // only the column part is loop invariant; the row part comes from the list, chunk by chunk
      a_voff[i] = ca < p.Cout ? (unsigned)(ca * (int)sizeof(T)) : kOOB;
      b_voff[i] = cb < p.Cin ? (unsigned)(cb * (int)sizeof(T)) : kOOB;
      m_voff[i] = (unsigned)(v * 8);
      const rowpair rp = __builtin_amdgcn_raw_buffer_load_b64(mr, m_voff[i], 0, 0);      // beyond the list: zeros = invalid row
      r_vox[i] = rp[0]; m_next[i] = rp[1];
      continue;
    }
    a_voff[i] = ca < p.Cout ? (unsigned)((v * p.Cout + ca) * (long long)sizeof(T)) : kOOB;
    b_voff[i] = cb < p.Cin ? (unsigned)((v * p.Cin + cb) * (long long)sizeof(T)) : kOOB;       // the tap shift is added per chunk
    m_voff[i] = (unsigned)(v * 4);
    m_next[i] = (v < p.M) ? 1u : 0u;
    if (MODE == 0 && use_mask) m_next[i] = __builtin_amdgcn_raw_buffer_load_b32(mr, m_voff[i], 0, 0);
  }
  const unsigned a_step = (unsigned)(KV * p.Cout * (int)sizeof(T)), b_step = (unsigned)(KV * p.Cin * (int)sizeof(T));
  const unsigned a_rowb = (unsigned)(p.Cout * (int)sizeof(T)), b_rowb = (unsigned)(p.Cin * (int)sizeof(T));
  // ROWS: addresses of piece i of the chunk whose list entries are in (r_vox, m_next); then fetch the next chunk's entries
  auto rows_addr = [&](int i, unsigned &aa, unsigned &bb) {
    const unsigned word = m_next[i];
    const bool valid = (word >> 13) & 1u;
    const bool in = p.taps == 27 ? ((word >> tap) & 1u) : valid;
    aa = (valid && a_voff[i] != kOOB) ? r_vox[i] * a_rowb + a_voff[i] : kOOB;
    bb = (in && b_voff[i] != kOOB) ? r_vox[i] * b_rowb + b_voff[i] + (unsigned)seg_shift[word >> 27] : kOOB;
    m_voff[i] += KV * 8;
    const rowpair rp = __builtin_amdgcn_raw_buffer_load_b64(mr, m_voff[i], 0, 0);
    r_vox[i] = rp[0]; m_next[i] = rp[1];
  };

  auto load_chunk = [&](long long ch) {
    const long long v0 = ch * KV;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      if (MODE == 0 && ROWS) {
        unsigned aa, bb;
        rows_addr(i, aa, bb);
        ra[i] = bufld16(dyr, aa);
        rb[i] = bufld16(xr, bb);
      } else if (MODE == 0) {
        ra[i] = bufld16(dyr, a_voff[i]);
        const bool in = (m_next[i] >> tap) & 1u;      // 0 beyond the tensor (the mask load itself was out of range)
        rb[i] = bufld16(xr, (in && b_voff[i] != kOOB) ? b_voff[i] + (unsigned)seg_shift[m_next[i] >> 27] : kOOB);
        a_voff[i] = a_voff[i] == kOOB ? kOOB : a_voff[i] + a_step;
        b_voff[i] = b_voff[i] == kOOB ? kOOB : b_voff[i] + b_step;
        m_voff[i] += KV * 4;
        if (use_mask) m_next[i] = __builtin_amdgcn_raw_buffer_load_b32(mr, m_voff[i], 0, 0);
        else m_next[i] = (m_voff[i] < (unsigned)(p.M * 4)) ? 1u : 0u;
      } else {
        const int pc = tid + 256 * i;
        const int row = pc / PIECES_ROW, col = pc % PIECES_ROW;   // voxel row in the chunk, 16-byte column
        const long long v = v0 + row;
        const bool vok = v < p.M;
        {
          const int c0 = m0 + col * (16 / (int)sizeof(T));
          ra[i] = bufld16(dyr, (vok && c0 < p.Cout) ? (unsigned)((v * p.Cout + c0) * (long long)sizeof(T)) : kOOB);
        }
        // stem: column = taps [t0, t0 + TPS) x 4 channels of the im2col row of output voxel v
        constexpr int TPS = 16 / (4 * (int)sizeof(T));
        f4 val = zero4();
        {
          const long long vv = vok ? v : 0;
          const int oz = (int)(vv % p.OZ);
          const long long t1 = vv / p.OZ;
          const int oy = (int)(t1 % p.OY);
          const long long t2 = t1 / p.OY;
          const int ox = (int)(t2 % p.OX);
          const long long nb = (t2 / p.OX) * (long long)p.X * p.Y * p.Z * 4;
#pragma unroll
          for (int q = 0; q < TPS; ++q) {
            const int tp = (n0 / 4) + col * TPS + q;
            const int tx = tp / 49, ty = (tp / 7) % 7, tz = tp % 7;
            const int ix = ox * p.stride - 3 + tx, iy = oy * p.stride - 3 + ty, iz = oz * p.stride - 3 + tz;
            const bool in = vok && tp < p.taps && (unsigned)ix < (unsigned)p.X && (unsigned)iy < (unsigned)p.Y && (unsigned)iz < (unsigned)p.Z;
            const unsigned off = in ? (unsigned)((nb + (((long long)ix * p.Y + iy) * p.Z + iz) * 4) * (long long)sizeof(T)) : kOOB;
            if (sizeof(T) == 4) val = bufld16(xr, off);
            else {
              typedef __attribute__((ext_vector_type(2))) unsigned int u2v;
              const u2v h2 = __builtin_amdgcn_raw_buffer_load_b64(xr, off, 0, 0);
              val[2 * q] = __uint_as_float(h2[0]); val[2 * q + 1] = __uint_as_float(h2[1]);
            }
          }
        }
        rb[i] = val;
      }
    }
  };

  // MODE 0: LDS-DMA straight into buffer `buf` (lane l of a wave-instruction lands at base + 16 l = physical slot order)
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_dma = [&](int buf) {
    char *A = lds + buf * 2 * TILE;
    char *B = A + TILE;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int dst = (64 * wave_u + 256 * i) * 16;
      if (ROWS) {
        unsigned aa, bb;
        rows_addr(i, aa, bb);
        lds_dma16(dyr, A + dst, aa);
        lds_dma16(xr, B + dst, bb);
        continue;
      }
      lds_dma16(dyr, A + dst, a_voff[i]);
      bool in = (m_next[i] >> tap) & 1u;
      unsigned shift = (unsigned)seg_shift[m_next[i] >> 27];
      if (PACK2 && b_half[i]) {
        in = tap1_ok && ((m_next[i] >> (tap + 1)) & 1u);
        shift = (unsigned)seg_shift1[m_next[i] >> 27];
      }
      lds_dma16(xr, B + dst, (in && b_voff[i] != kOOB) ? b_voff[i] + shift : kOOB);
      a_voff[i] = a_voff[i] == kOOB ? kOOB : a_voff[i] + a_step;
      b_voff[i] = b_voff[i] == kOOB ? kOOB : b_voff[i] + b_step;
      m_voff[i] += KV * 4;
      if (use_mask) m_next[i] = __builtin_amdgcn_raw_buffer_load_b32(mr, m_voff[i], 0, 0);
      else m_next[i] = (m_voff[i] < (unsigned)(p.M * 4)) ? 1u : 0u;
    }
  };

  auto store_chunk = [&](int buf) {   // register-staged path (stem gather): MODE-1 pieces are in logical column order
    char *A = lds + buf * 2 * TILE;
    char *B = A + TILE;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int pc = tid + 256 * i;
      const int row = pc / PIECES_ROW, col = pc % PIECES_ROW;
      *reinterpret_cast<f4 *>(A + wg_off(row, col * 16, RS)) = ra[i];
      *reinterpret_cast<f4 *>(B + wg_off(row, col * 16, RS)) = rb[i];
    }
  };

  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_bias = p.gbias != nullptr && n0 == 0 && tap == (PACK2 ? 12 : ((MODE == 0 && p.taps == 27) ? 13 : 0));      // PACK2: the pair (12, 13)
  float bias_acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) issue_dma(0);
  else { load_chunk(c_begin); store_chunk(0); }
  __syncthreads();
  int buf = 0;
  for (long long ch = c_begin; ch < c_end; ++ch) {
    if (ch + 1 < c_end) { if (MODE == 0) issue_dma(buf ^ 1); else load_chunk(ch + 1); }
    const char *A = lds + buf * 2 * TILE;
    const char *B = A + TILE;
    if (do_bias) {   // thread t owns one 16-byte column group of the dY tile and every (256 / PIECES_ROW)-th voxel row
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const f4 v = *reinterpret_cast<const f4 *>(A + (tid / PIECES_ROW + i * (256 / PIECES_ROW)) * RS + (tid % PIECES_ROW) * 16);   // physical slot
        if (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bias_acc[e] += v[e];
        } else {
          typedef __attribute__((ext_vector_type(8))) unsigned short u8v;
          const u8v h = __builtin_bit_cast(u8v, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) bias_acc[e] += bf16_bits_to_f32(h[e]);
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < KV; kb += KSUB) {
      f4 af[2], bfv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = wg_frag<T, TR>(A, (wm * 2 + i) * 32, kb, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfv[j] = wg_frag<T, TR>(B, (wn * 2 + j) * 32, kb, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<T>::run(acc[i][j], af[i], bfv[j]);
    }
    if (MODE != 0 && ch + 1 < c_end) store_chunk(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  if (do_bias) {   // reduce the row groups through LDS (free after the loop's last barrier): one partial per column per workgroup
    constexpr int EPP = 16 / (int)sizeof(T), NRG = 256 / PIECES_ROW;
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int e = 0; e < EPP; ++e) red[tid * 8 + e] = bias_acc[e];
    __syncthreads();
    if (tid < 128) {
      const int g = tid / EPP, e = tid % EPP;
      float sum = 0.f;
#pragma unroll
      for (int rg = 0; rg < NRG; ++rg) sum += red[(rg * PIECES_ROW + (g ^ ((rg & 3) << 2))) * 8 + e];   // thread holding logical slot g of row group rg
      if (m0 + tid < p.Cout) p.gbias[(long long)slice * p.wrows + m0 + tid] = sum;
    }
  }
  const int fr = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int col = n0 + (wn * 2 + j) * 32 + fr;
    const int ncols = (MODE == 0) ? p.Cin : p.kpad;
    int otap = tap;
    if (PACK2) {        // tile columns [0, 64) = the pair's first tap, [64, 128) = its second: wave column wn IS the tap of the pair
      otap = tap + (col >> 6);
      col &= 63;
      if (otap >= 27) continue;
    }
    if (col >= ncols) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * 2 + i) * 32 + frag_row(r, lane);
        if (row < p.wrows) {   // plain store into this slice's partial gradient (summed by nrpn_unpack_*_wgrad): no atomics, no memset
          float *dst = p.gw + slice * p.slice_stride +
                       ((MODE == 0) ? ((long long)otap * p.wrows + row) * p.Cin + col : (long long)row * p.kpad + col);
          *dst = acc[i][j][r];
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256(Cout) x 256(Cin) wgrad tile for the wide bf16 layers: 8 waves (2 x 4, each 128x64), four [64 voxel][128 channel]
// sub-tiles per LDS buffer in the layout of the 128-wide kernel (so wg_frag / the transpose reads are unchanged).  Halves the
// bytes pulled through the CU's vector-memory path per MFMA (the 128x128 tile moves 32 KB per 16 MFMAs per wave and saturates
// it).  One workgroup = (cout tile, cin tile, tap, voxel slice); fp32 atomics into the packed gradient.
// ---------------------------------------------------------------------------------------------------------------------
// PACK2 (round 5, dense k3, Cin == 128, Cout >= 256: layers.5.0 of VGG19): the two B sub-tiles hold the SAME 128 input channels of a PAIR of
// taps (sub-tile t = tap 2s + t) instead of channels [0, 128) / [128, 256) of one tap -- the layer ran on the 128 x 128 kernel at 580 TFLOP/s
// because half of a 256-column tile would have been empty.  14 workgroups per (tile, slice); tile columns [128 t, 128 t + 128) = tap 2s + t.
template <bool ROWS, bool PACK2 = false>
__global__ void __launch_bounds__(512, 1) conv_wgrad_big_kernel(const WgradArgs p) {
  static_assert(!(ROWS && PACK2), "PACK2 is a dense form");
  typedef bf16s T;
  constexpr int KV = 64, RS = 256, SUB = KV * RS;            // 16 KB sub-tile
  constexpr int PIECES_ROW = 16, KSUB = 16, TM = 4, TN = 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];   // [2 buffers][A0 | A1 | B0 | B1]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned ntm = (p.wrows + 255) / 256, ntn = p.ntiles_n, tps = PACK2 ? 14u : (unsigned)p.taps;
  unsigned id = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (int)(id % ntm) * 256;
  id /= ntm;
  const int n0 = (int)(id % ntn) * 256;
  id /= ntn;
  const int tap = PACK2 ? 2 * (int)(id % tps) : (int)(id % tps);       // PACK2: the pair's first tap
  const int slice = (int)(id / tps);
  int dx = 0, dy = 0, dz = 0;
  if (p.taps == 27) { dx = tap / 9 - 1; dy = (tap / 3) % 3 - 1; dz = tap % 3 - 1; }
  const bool tap1_ok = PACK2 && tap + 1 < 27;
  const long long chunks = (p.M + KV - 1) / KV;
  const long long per = (chunks + p.ksplit - 1) / p.ksplit;
  const long long c_begin = slice * per, c_end = min(chunks, c_begin + per);
  if (c_begin >= c_end) return;

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), dyr = make_rsrc(p.dy, p.dy_bytes);
  const __amdgpu_buffer_rsrc_t mr = ROWS ? make_rsrc(p.rows, (unsigned)(p.M * 8)) : make_rsrc(p.vmask, (unsigned)(p.M * 4));
  __shared__ int seg_shift[kMaxSeg];            // byte shift of this tap inside each segment's grid (see conv_wgrad_kernel)
  __shared__ int seg_shift1[PACK2 ? kMaxSeg : 1];       // PACK2: the pair's second tap
  if (tid < kMaxSeg) {
    int Y = p.Y, Z = p.Z;
#pragma unroll
    for (int q = 0; q < kMaxSeg; ++q)
      if (q == tid && q < p.segs.n) { Y = p.segs.Y[q]; Z = p.segs.Z[q]; }
    seg_shift[tid] = (int)((((long long)dx * Y + dy) * Z + dz) * p.Cin * 2);
    if (PACK2) {
      const int t1 = tap + 1, dx1 = t1 / 9 - 1, dy1 = (t1 / 3) % 3 - 1, dz1 = t1 % 3 - 1;
      seg_shift1[tid] = (int)((((long long)dx1 * Y + dy1) * Z + dz1) * p.Cin * 2);
    }
  }
  __syncthreads();
  const bool use_mask = p.taps == 27;
  // per thread: rows r_i = tid / 16 + 32 i (i = 0, 1) of every sub-tile, physical 16-byte slot tid % 16
  unsigned a_voff[2][2], b_voff[2][2], m_voff[2], m_next[2];
  unsigned r_vox[2] = {0u, 0u};   // ROWS: voxel id of this thread's row in the next chunk (fetched with its tap word)
  typedef __attribute__((ext_vector_type(2))) unsigned int rowpair;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = tid / PIECES_ROW + 32 * i;
    const int col = (tid % PIECES_ROW) ^ ((row & 3) << 2);          // logical column of this physical slot
    const long long v = c_begin * KV + row;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int ca = m0 + 128 * t + col * 8, cb = PACK2 ? col * 8 : n0 + 128 * t + col * 8;       // PACK2: both B sub-tiles start at channel 0
      if (ROWS) {     // only the column part is loop invariant; the row part comes from the list, chunk by chunk
        a_voff[t][i] = ca < p.Cout ? (unsigned)(ca * 2) : kOOB;
        b_voff[t][i] = cb < p.Cin ? (unsigned)(cb * 2) : kOOB;
      } else {
        a_voff[t][i] = ca < p.Cout ? (unsigned)((v * p.Cout + ca) * 2) : kOOB;
        b_voff[t][i] = cb < p.Cin ? (unsigned)((v * p.Cin + cb) * 2) : kOOB;                   // the tap shift is added per chunk
      }
    }
    if (ROWS) {
      m_voff[i] = (unsigned)(v * 8);
      const rowpair rp = __builtin_amdgcn_raw_buffer_load_b64(mr, m_voff[i], 0, 0);           // beyond the list: zeros = invalid row
      r_vox[i] = rp[0]; m_next[i] = rp[1];
      continue;
    }
    m_voff[i] = (unsigned)(v * 4);
    m_next[i] = (v < p.M) ? 1u : 0u;
    if (use_mask) m_next[i] = __builtin_amdgcn_raw_buffer_load_b32(mr, m_voff[i], 0, 0);
  }
  const unsigned a_rowb = (unsigned)(p.Cout * 2), b_rowb = (unsigned)(p.Cin * 2);
  const unsigned a_step = (unsigned)(KV * p.Cout * 2), b_step = (unsigned)(KV * p.Cin * 2);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue_dma = [&](int buf) {
    char *base = lds + buf * 4 * SUB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int dst = (64 * wave_u + 512 * i) * 16;
      if (ROWS) {
        const unsigned word = m_next[i];
        const bool valid = (word >> 13) & 1u;
        const bool inr = p.taps == 27 ? ((word >> tap) & 1u) : valid;
        const unsigned ra = r_vox[i] * a_rowb, rb = r_vox[i] * b_rowb + (unsigned)seg_shift[word >> 27];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          lds_dma16(dyr, base + t * SUB + dst, (valid && a_voff[t][i] != kOOB) ? ra + a_voff[t][i] : kOOB);
          lds_dma16(xr, base + (2 + t) * SUB + dst, (inr && b_voff[t][i] != kOOB) ? rb + b_voff[t][i] : kOOB);
        }
        m_voff[i] += KV * 8;
        const rowpair rp = __builtin_amdgcn_raw_buffer_load_b64(mr, m_voff[i], 0, 0);
        r_vox[i] = rp[0]; m_next[i] = rp[1];
        continue;
      }
      const bool in0 = (m_next[i] >> tap) & 1u;
      const unsigned tsh0 = (unsigned)seg_shift[m_next[i] >> 27];
      const bool in1 = PACK2 ? (tap1_ok && ((m_next[i] >> (tap + 1)) & 1u)) : in0;
      const unsigned tsh1 = PACK2 ? (unsigned)seg_shift1[m_next[i] >> 27] : tsh0;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool in = t ? in1 : in0;
        const unsigned tsh = t ? tsh1 : tsh0;
        lds_dma16(dyr, base + t * SUB + dst, a_voff[t][i]);
        lds_dma16(xr, base + (2 + t) * SUB + dst, (in && b_voff[t][i] != kOOB) ? b_voff[t][i] + tsh : kOOB);
        a_voff[t][i] = a_voff[t][i] == kOOB ? kOOB : a_voff[t][i] + a_step;
        b_voff[t][i] = b_voff[t][i] == kOOB ? kOOB : b_voff[t][i] + b_step;
      }
      m_voff[i] += KV * 4;
      if (use_mask) m_next[i] = __builtin_amdgcn_raw_buffer_load_b32(mr, m_voff[i], 0, 0);
      else m_next[i] = (m_voff[i] < (unsigned)(p.M * 4)) ? 1u : 0u;
    }
  };

  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool do_bias = p.gbias != nullptr && n0 == 0 && tap == (PACK2 ? 12 : (p.taps == 27 ? 13 : 0));       // PACK2: the pair (12, 13)
  float bias_acc[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) bias_acc[t][e] = 0.f;

  issue_dma(0);
  __syncthreads();
  int buf = 0;
  // Rotated pipeline (cf. conv_igemm_big_kernel): the barrier that hands over the next chunk sits before the MFMAs of the LAST sub-step,
  // whose fragments are already in registers; those MFMAs then cover the transpose reads of the next chunk's sub-step 0.  Same MFMA
  // order per accumulator as the plain loop.  (Issuing the DMA of waves 4-7 later in the chunk, as the forward kernel does, was
  // measured 10 % SLOWER here: 208 -> 229 us.)
  constexpr int NSUB = KV / KSUB;
  static_assert(NSUB % 2 == 0, "fragment double-buffer parity across chunks");
  f4 af[2][TM], bfv[2][TN];
  {
    const char *A0 = lds + wm * SUB, *B0 = lds + (2 + (wn >> 1)) * SUB;
#pragma unroll
    for (int j = 0; j < TN; ++j) bfv[0][j] = wg_frag<T, true>(B0, (wn & 1) * 64 + j * 32, 0, lane);
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = wg_frag<T, true>(A0, i * 32, 0, lane);
  }
#pragma unroll 1
  for (long long ch = c_begin; ch < c_end; ++ch) {
    if (ch + 1 < c_end) issue_dma(buf ^ 1);
    const char *base = lds + buf * 4 * SUB;
    const char *A = base + wm * SUB;
    const char *B = base + (2 + (wn >> 1)) * SUB;
    const char *nbase = lds + (buf ^ 1) * 4 * SUB;
    if (do_bias) {
      typedef __attribute__((ext_vector_type(8))) unsigned short u8v;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f4 v = *reinterpret_cast<const f4 *>(base + t * SUB + (tid / PIECES_ROW + 32 * i) * RS + (tid % PIECES_ROW) * 16);
          const u8v h = __builtin_bit_cast(u8v, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) bias_acc[t][e] += bf16_bits_to_f32(h[e]);
        }
    }
    // register double-buffered fragments: the 12 transpose reads of sub-step q+1 ride behind the first six MFMAs of sub-step q
#pragma unroll
    for (int q = 0; q < NSUB; ++q) {
      if (q + 1 < NSUB) {
#pragma unroll
        for (int j = 0; j < TN; ++j) bfv[(q + 1) & 1][j] = wg_frag<T, true>(B, (wn & 1) * 64 + j * 32, (q + 1) * KSUB, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[(q + 1) & 1][i] = wg_frag<T, true>(A, i * 32, (q + 1) * KSUB, lane);
      } else {
        __syncthreads();          // every wave has read its last fragments of this chunk; the next chunk has landed
#pragma unroll
        for (int j = 0; j < TN; ++j) bfv[0][j] = wg_frag<T, true>(nbase + (2 + (wn >> 1)) * SUB, (wn & 1) * 64 + j * 32, 0, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i) af[0][i] = wg_frag<T, true>(nbase + wm * SUB, i * 32, 0, lane);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[q & 1][i], bfv[q & 1][j]);
#pragma unroll
      for (int g = 0; g < TM + TN; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
    }
    buf ^= 1;
  }
  __syncthreads();      // the bias reduction below reuses the LDS

  if (do_bias) {   // reduce the 32 row groups through LDS: one partial per column per workgroup
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(tid * 2 + t) * 8 + e] = bias_acc[t][e];
    __syncthreads();
    if (tid < 256) {
      const int t = tid >> 7, g = (tid & 127) >> 3, e = tid & 7;
      float sum = 0.f;
#pragma unroll 8
      for (int rg = 0; rg < 32; ++rg) sum += red[((rg * PIECES_ROW + (g ^ ((rg & 3) << 2))) * 2 + t) * 8 + e];
      if (m0 + tid < p.Cout) p.gbias[(long long)slice * p.wrows + m0 + tid] = sum;
    }
  }
  int elane = lane;
  asm volatile("" : "+v"(elane));
  const int fr = elane & 31;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int col = n0 + wn * 64 + j * 32 + fr;
    int otap = tap;
    if (PACK2) {        // tile columns [0, 128) = the pair's first tap, [128, 256) = its second
      otap = tap + (col >> 7);
      col &= 127;
      if (otap >= 27) continue;
    }
    if (col >= p.Cin) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 128 + i * 32 + frag_row(r, elane);
        if (row < p.wrows) p.gw[slice * p.slice_stride + ((long long)otap * p.wrows + row) * p.Cin + col] = acc[i][j][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem wgrad, stride 2, even Z ("z-row" form).  For a fixed (dx, dy) the 7 dz taps x 4 channels of the stem's 7^3 window are 28
// CONTIGUOUS elements of the input row (z fastest, then channel), and with stride 2 the row of output voxel oz starts at the even
// z = 2 oz - 4 when one unused leading z position is included: 8 z x 4 c = 32 elements = one 64-byte (bf16) / 128-byte (fp32)
// run that LDS-DMA can stage like an ordinary channel row.  The wgrad becomes 49 independent GEMMs
//     dWz[(dx,dy)][cout][zrel*4 + c] = sum_v dY[v][cout] * X[n, 2ox-3+dx, 2oy-3+dy, 2oz-4+zrel, c]        (zrel = dz + 1; zrel 0 unused)
// with M = Cout = 64, N = 32, K = voxels -- the structure of conv_wgrad_kernel (LDS tiles [voxel][channel], transpose reads), instead
// of an im2col gather of 8-byte pieces with per-piece tap decoding (the old path: 141 TFLOP/s, half of its 128-row M tile empty).
// Workgroup = (dx, voxel slice): the dY tile of a chunk is staged once and shared by the 7 dy taps, wave w owns taps {w, w+4}.
// Whole 16-byte pieces are either inside the grid or entirely outside (Z even => z pairs never straddle the border), so border
// handling is the buffer descriptor's zero fill.  Partials: gw[slice][dx*7+dy][cout][32] (plain stores), summed by the unpack.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct ZrCfg;
template <> struct ZrCfg<bf16s> { static constexpr int KV = 64, RSA = 64 * 2, RSB = 32 * 2, KSUB = 16; };
template <> struct ZrCfg<float> { static constexpr int KV = 32, RSA = 64 * 4, RSB = 32 * 4, KSUB = 8; };

// 16-byte-slot swizzle of a [voxel][channel] tile with RS-byte rows: keeps the rows one transpose-read group touches on distinct banks
template <int RS> __device__ __forceinline__ int zr_swz(int row) {
  if (RS >= 256) return (row & 3) << 2;
  if (RS == 128) return ((row >> 1) & 1) << 2;
  return 0;
}
template <int RS> __device__ __forceinline__ int zr_off(int row, int byte_in_row) {
  return row * RS + ((((byte_in_row >> 4) ^ zr_swz<RS>(row))) << 4) + (byte_in_row & 15);
}

// 32(channel) x 16-byte K fragment out of a [voxel][channel] tile with RS-byte rows (cf. wg_frag)
template <typename T, int RS>
__device__ __forceinline__ f4 zr_frag(const char *tile, int ctile0, int kbase, int lane) {
  const int h = lane >> 5;
  if (sizeof(T) == 4) {
    const int c = ctile0 + (lane & 31);
    f4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float *>(tile + zr_off<RS>(kbase + 4 * h + q, c * 4));
    return v;
  } else {
    const int p = lane & 15;
    const int cbase = ctile0 + 16 * ((lane >> 4) & 1);
    const int row = kbase + 8 * h + (p >> 2);              // row + 4 has the same swizzle term: one offset serves both reads
    const char *a0 = tile + zr_off<RS>(row, (cbase + 4 * (p & 3)) * 2);
    const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0));
    const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3))) *)(a0 + 4 * RS));
    typedef __attribute__((ext_vector_type(2))) long long l2v;
    l2v r = {__builtin_bit_cast(long long, lo), __builtin_bit_cast(long long, hi)};
    return __builtin_bit_cast(f4, r);
  }
}

struct StemZrArgs {
  const void *x, *dy;
  float *gw;            // [slices][49][64][32]
  float *gbias;         // optional per-slice bias partials [slices][64]
  long long M;          // output voxels
  int X, Y, Z, OX, OY, OZ;
  int slices;
  unsigned x_bytes, dy_bytes;
};

template <typename T>
__global__ void __launch_bounds__(256, 2) stem_wgrad_zrow_kernel(const StemZrArgs p) {
  constexpr int KV = ZrCfg<T>::KV, RSA = ZrCfg<T>::RSA, RSB = ZrCfg<T>::RSB, KSUB = ZrCfg<T>::KSUB;
  constexpr int COUT = 64;
  constexpr int A_BYTES = KV * RSA, B_BYTES = KV * RSB, BUF = A_BYTES + 7 * B_BYTES;
  constexpr int APR = RSA / 16, BPR = RSB / 16;                 // 16-byte pieces per row: 8 / 4 (bf16), 16 / 8 (fp32)
  constexpr int A_PIECES = KV * APR / 256;                      // per lane per chunk: 2
  static_assert(KV * BPR == 256, "one LDS-DMA instruction per tap");
  extern __shared__ __attribute__((aligned(16))) char lds[];    // [2 buffers][dY tile | 7 X-row tiles]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dx = blockIdx.x % 7, slice = blockIdx.x / 7;
  const long long chunks = (p.M + KV - 1) / KV;
  const long long per = (chunks + p.slices - 1) / p.slices;
  const long long c_begin = slice * per, c_end = min(chunks, c_begin + per);

  const __amdgpu_buffer_rsrc_t xr = make_rsrc(p.x, p.x_bytes), dyr = make_rsrc(p.dy, p.dy_bytes);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // A (dY) pieces of this lane: physical slot tid % APR of rows tid / APR + (256 / APR) * i, holding logical slot (slot ^ swz(row))
  unsigned a_voff[A_PIECES];
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int pc = tid + 256 * i, row = pc / APR, slot = (pc % APR) ^ zr_swz<RSA>(row);
    a_voff[i] = (unsigned)(((c_begin * KV + row) * COUT) * (long long)sizeof(T)) + slot * 16;
  }
  const unsigned a_step = (unsigned)(KV * COUT * (int)sizeof(T));
  const int b_row = tid / BPR, b_slot = (tid % BPR) ^ zr_swz<RSB>(tid / BPR);   // X-row piece of this lane (logical slot; the same for every dy tap)
  constexpr int ZPP = 16 / (4 * (int)sizeof(T));                // z positions per 16-byte piece: 2 (bf16) / 1 (fp32)
  // Output voxel of this lane's X-row piece, v = chunk * KV + b_row: decomposed ONCE and advanced by KV per chunk.  (Three 64-bit
  // divisions per lane and chunk were ~350 VALU instructions next to the chunk's 16 MFMAs: the kernel was issue-bound on index arithmetic.)
  long long b_v = c_begin * KV + b_row;
  int b_oz, b_oy, b_ox, b_nb;
  {
    const long long q1 = b_v / p.OZ, q2 = q1 / p.OY, q3 = q2 / p.OX;
    b_oz = (int)(b_v - q1 * p.OZ); b_oy = (int)(q1 - q2 * p.OY); b_ox = (int)(q2 - q3 * p.OX); b_nb = (int)q3;
  }

  auto issue = [&](int buf, long long ch) {
    char *A = lds + buf * BUF;
    char *B = A + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const long long v = ch * KV + (tid + 256 * i) / APR;
      lds_dma16(dyr, A + (64 * wave_u + 256 * i) * 16, v < p.M ? a_voff[i] : kOOB);
      a_voff[i] += a_step;
    }
    // (issue() is called for consecutive chunks, starting at c_begin: b_v / b_oz.. are this chunk's, then advance)
    const bool vok = b_v < p.M;
    const int ix = 2 * b_ox - 3 + dx;
    const int zp = 2 * b_oz - 4 + b_slot * ZPP;                 // first z of this piece (ZPP z positions, never straddling the border: Z even)
    const bool ok0 = vok && (unsigned)ix < (unsigned)p.X && zp >= 0 && zp + ZPP <= p.Z;
    const int rowbase = ((b_nb * p.X + ix) * p.Y) * p.Z;        // voxel index: fits 32 bits (x_bytes is a 32-bit buffer size); unused when !ok0
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int iy = 2 * b_oy - 3 + t;
      const bool ok = ok0 && (unsigned)iy < (unsigned)p.Y;
      const unsigned off = ok ? (unsigned)(rowbase + iy * p.Z + zp) * (unsigned)(4 * sizeof(T)) : kOOB;
      lds_dma16(xr, B + t * B_BYTES + 64 * wave_u * 16, off);
    }
    b_v += KV;
    b_oz += KV;
    while (b_oz >= p.OZ) {
      b_oz -= p.OZ;
      if (++b_oy == p.OY) {
        b_oy = 0;
        if (++b_ox == p.OX) { b_ox = 0; ++b_nb; }
      }
    }
  };

  const int wm_tiles = 2;                                        // 64 couts = 2 row tiles of 32
  const int t0 = wave, t1 = wave + 4;                            // dy taps of this wave (t1 < 7 for waves 0..2)
  const bool two = t1 < 7;
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const bool do_bias = p.gbias != nullptr && dx == 0;
  float bias_acc[16 / (int)sizeof(T)];
#pragma unroll
  for (int e = 0; e < 16 / (int)sizeof(T); ++e) bias_acc[e] = 0.f;

  if (c_begin < c_end) issue(0, c_begin);
  __syncthreads();
  int buf = 0;
  for (long long ch = c_begin; ch < c_end; ++ch) {
    if (ch + 1 < c_end) issue(buf ^ 1, ch + 1);
    const char *A = lds + buf * BUF;
    const char *B = A + A_BYTES;
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i) {
        const f4 v = *reinterpret_cast<const f4 *>(A + (tid + 256 * i) * 16);      // this lane's own piece (physical slot order)
        if (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) bias_acc[e] += v[e];
        } else {
          typedef __attribute__((ext_vector_type(8))) unsigned short u8v;
          const u8v hh = __builtin_bit_cast(u8v, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) bias_acc[e] += bf16_bits_to_f32(hh[e]);
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < KV; kb += KSUB) {
      f4 af[2], bf0, bf1;
#pragma unroll
      for (int i = 0; i < wm_tiles; ++i) af[i] = zr_frag<T, RSA>(A, i * 32, kb, lane);
      bf0 = zr_frag<T, RSB>(B + t0 * B_BYTES, 0, kb, lane);
      if (two) bf1 = zr_frag<T, RSB>(B + t1 * B_BYTES, 0, kb, lane);
#pragma unroll
      for (int i = 0; i < wm_tiles; ++i) {
        Mma<T>::run(acc[i][0], af[i], bf0);
        if (two) Mma<T>::run(acc[i][1], af[i], bf1);
      }
    }
    __syncthreads();
    buf ^= 1;
  }

  if (do_bias) {   // thread t summed logical slot (t % APR) ^ swz(row) of rows t / APR + k * (256 / APR); reduce the row groups through LDS
    constexpr int EPP = 16 / (int)sizeof(T), NRG = 256 / APR;
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int e = 0; e < EPP; ++e) red[tid * EPP + e] = bias_acc[e];
    __syncthreads();
    if (tid < COUT) {
      const int g = tid / EPP, e = tid % EPP;
      float sum = 0.f;
      for (int rg = 0; rg < NRG; ++rg) sum += red[(rg * APR + (g ^ zr_swz<RSA>(rg))) * EPP + e];
      p.gbias[(long long)slice * COUT + tid] = sum;
    }
  }
  const int fr = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j == 1 && !two) continue;
    const int tap = dx * 7 + (j == 0 ? t0 : t1);
    float *dst = p.gw + ((long long)slice * 49 + tap) * COUT * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[(i * 32 + frag_row(r, lane)) * 32 + fr] = acc[i][j][r];
  }
}

// z-row partials [slices][49][cout][32] -> reference layout [cout][4][7][7][7] (summed over the slices; zrel 0 is the unused position)
__global__ void unpack_stem_zrow_kernel(const float *__restrict__ gp, int cout, float *__restrict__ gw, int accumulate, int slices) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)cout * 4 * 343) return;
  const int tap = (int)(i % 343), c = (int)((i / 343) % 4), o = (int)(i / (343 * 4));
  const int dz = tap % 7, dxy = tap / 7;
  const float *src = gp + ((long long)dxy * cout + o) * 32 + (dz + 1) * 4 + c;
  // four independent chains (fixed order: deterministic): the loop is latency-bound, 73 slices x one strided float per thread
  const long long st = (long long)49 * cout * 32;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  int s = 0;
  for (; s + 3 < slices; s += 4) { v0 += src[s * st]; v1 += src[(s + 1) * st]; v2 += src[(s + 2) * st]; v3 += src[(s + 3) * st]; }
  for (; s < slices; ++s) v0 += src[s * st];
  const float v = (v0 + v1) + (v2 + v3);
  gw[i] = accumulate ? gw[i] + v : v;
}

static bool stem_zrow_ok(int gz, int cout, int stride) { return stride == 2 && gz % 2 == 0 && cout == 64; }
static int stem_zrow_slices(long long M, int elem_bytes) {
  const long long chunks = (M + (elem_bytes == 2 ? 64 : 32) - 1) / (elem_bytes == 2 ? 64 : 32);
  long long s = 73;                                   // 7 dx x 73 slices = 511 workgroups = two per CU
  if (s > chunks / 8) s = chunks / 8;
  if (s < 1) s = 1;
  const long long per = (chunks + s - 1) / s;
  return (int)((chunks + per - 1) / per);             // no empty slice
}

// bias gradient = ordered sum of the per-slice column sums the wgrad workgroups left in the workspace
__global__ void bias_finalize_kernel(const float *__restrict__ part, int slices, int wrows, int cout, float *__restrict__ gbias, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cout) return;
  // eight independent chains in a fixed order (deterministic): one float per slice and lane, the serial loop over 73 stem slices was 26 us
  // of pure load latency at the tail of backward
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 7 < slices; k += 8)
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += part[(long long)(k + u) * wrows + c];
  for (; k < slices; ++k) a[0] += part[(long long)k * wrows + c];
  const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  gbias[c] = accumulate ? gbias[c] + s : s;
}

// bit t of mask[v] = 1 iff voxel v shifted by tap t = (dx+1)*9 + (dy+1)*3 + (dz+1) stays inside its own grid; bits 27..31 = the
// voxel's segment id (0 in the classic layout)
__global__ void tap_mask_kernel(unsigned *__restrict__ mask, long long M, int X, int Y, int Z, Segs segs) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= M) return;
  int x, y, z, gX, gY, gZ;
  const int seg = locate_voxel(segs, v, X, Y, Z, x, y, z, gX, gY, gZ);
  unsigned m = (unsigned)seg << 27;
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const int dx = t / 9 - 1, dy = (t / 3) % 3 - 1, dz = t % 3 - 1;
    if ((unsigned)(x + dx) < (unsigned)gX && (unsigned)(y + dy) < (unsigned)gY && (unsigned)(z + dz) < (unsigned)gZ) m |= 1u << t;
  }
  mask[v] = m;
}

static std::atomic<int> g_wgrad_tr_mode{1};
extern "C" int nrpn_set_wgrad_transpose_read(int on) { g_wgrad_tr_mode = on ? 1 : 0; return NRPN_OK; }

static std::atomic<int> g_wgrad_big{1};   // 1: 256x256 wgrad tiles for bf16 layers with Cout, Cin >= 256 (tuning knob)
extern "C" int nrpn_set_wgrad_big_tile(int on) { g_wgrad_big = on ? 1 : 0; return NRPN_OK; }

// How the voxel axis is cut: every (tile, tap, slice) workgroup writes one partial gradient; the slices are summed by the
// unpack kernels.  `ksplit` is trimmed so that no slice is empty (every partial is fully written).
// two taps per workgroup (conv_wgrad_kernel<..., PACK2>): dense k3 layers whose input row fits half a B-tile row (Cin <= 64)
static std::atomic<int> g_wgrad_pack2{1};
extern "C" int nrpn_set_wgrad_pack2(int on) { g_wgrad_pack2 = on ? 1 : 0; return NRPN_OK; }
static bool wgrad_pack2(int cin, int taps, int mode) { return g_wgrad_pack2.load(std::memory_order_relaxed) && mode == 0 && taps == 27 && cin <= 64; }
struct WgPlan { int ksplit; bool big; };
static WgPlan wgrad_plan(long long M, int wrows, int ncols, int taps, int elem_bytes, int mode, int cout, int cin) {
  WgPlan pl{1, false};
  const int kv = elem_bytes == 4 ? 32 : 64;
  const long long chunks = (M + kv - 1) / kv;
  long long ks = 1;
  const bool big_pack2 = g_wgrad_pack2.load(std::memory_order_relaxed) && taps == 27 && cin == 128;      // conv_wgrad_big_kernel<false, PACK2>
  if (mode == 0 && elem_bytes == 2 && g_wgrad_big && g_wgrad_tr_mode != 0 && cout >= 256 && (cin >= 256 || big_pack2)) {
    // 256x256 tiles, one 128 KB-LDS workgroup per CU: pick the slice count with the lowest modelled time
    //   t(k) = MFMA time / (fill of whole rounds of 256 CUs) + (k + 1) partial-gradient passes through HBM,
    // e.g. 256->256 @40^3: 27 tiles x 9 slices = 243 workgroups (95 % of a round) instead of 8 (84 %); 512->512 @20^3 stays at 2
    // slices because each extra slice costs another 28 MB partial.
    const int tiles = ((wrows + 255) / 256) * ((cin + 255) / 256) * ((big_pack2 && cin == 128) ? 14 : taps);
    const double flops = 2.0 * (double)M * wrows * cin * taps;
    const double part_bytes = 4.0 * (double)taps * wrows * cin;
    double best = 1e30;
    for (long long k = 1; k <= 64 && k <= max(1ll, chunks / 16); ++k) {
      const double rounds = (double)tiles * k / 256.0;
      const long long whole = (long long)rounds + ((double)(long long)rounds < rounds ? 1 : 0);
      const double fill = rounds / (double)whole;
      if (fill < 0.5) continue;
      const double t = flops / (fill * 9.0e14) + (double)(k + 1) * part_bytes / 3.0e12;
      if (t < best) { best = t; pl.big = true; ks = k; }
    }
  }
  if (!pl.big) {
    const int tiles = ((wrows + 127) / 128) * ((ncols + 127) / 128) * (wgrad_pack2(cin, taps, mode) ? 14 : taps);
    ks = (1024 + tiles - 1) / tiles;
    if (ks > chunks / 16) ks = chunks / 16;     // >= 16 chunks per workgroup so the 128x128 epilogue stays amortised
    if (ks < 1) ks = 1;
    if (ks > 4096) ks = 4096;
  }
  const long long per = (chunks + ks - 1) / ks;
  pl.ksplit = (int)((chunks + per - 1) / per);
  return pl;
}

template <typename T, int MODE>
static int launch_wgrad(WgradArgs a, int ntiles_n, hipStream_t st) {
  a.ntiles_n = ntiles_n;
  const size_t lds = 4 * (size_t)WgCfg<T>::KV * WgCfg<T>::RS;
  const bool tr = g_wgrad_tr_mode != 0;
  dim3 grid((unsigned)(((a.wrows + 127) / 128) * ntiles_n * (MODE == 0 ? a.taps : 1) * a.ksplit));
  if constexpr (MODE == 0) {
    if (a.rows) {      // row-list form (transpose-read fragments only)
      NRPN_LDS((conv_wgrad_kernel<T, 0, true, true>), (int)lds);
      hipLaunchKernelGGL((conv_wgrad_kernel<T, 0, true, true>), grid, dim3(256), lds, st, a);
      NRPN_LAUNCH_CHECK("conv_wgrad_rows");
      return NRPN_OK;
    }
  }
  if constexpr (MODE == 0) {
    if (tr && !a.rows && wgrad_pack2(a.Cin, a.taps, 0)) {        // two taps per workgroup: 14 instead of 27 workgroups per (tile, slice)
      grid = dim3((unsigned)(((a.wrows + 127) / 128) * 14 * a.ksplit));
      a.ntiles_n = 1;
      NRPN_LDS((conv_wgrad_kernel<T, 0, true, false, true>), (int)lds);
      hipLaunchKernelGGL((conv_wgrad_kernel<T, 0, true, false, true>), grid, dim3(256), lds, st, a);
      NRPN_LAUNCH_CHECK("conv_wgrad_pack2");
      return NRPN_OK;
    }
  }
  if (tr) NRPN_LDS((conv_wgrad_kernel<T, MODE, true>), (int)lds);
  else NRPN_LDS((conv_wgrad_kernel<T, MODE, false>), (int)lds);
  if (tr) hipLaunchKernelGGL((conv_wgrad_kernel<T, MODE, true>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((conv_wgrad_kernel<T, MODE, false>), grid, dim3(256), lds, st, a);
  NRPN_LAUNCH_CHECK("conv_wgrad");
  return NRPN_OK;
}

// workspace = [tap mask: M u32 (k3 only)][bias partials: slices * wrows f32]
static size_t wgrad_mask_bytes(long long M, int ksize) { return ksize == 3 ? (size_t)((M * 4 + 255) / 256 * 256) : 0; }
extern "C" size_t nrpn_conv3d_wgrad_bias_offset(int n, int gx, int gy, int gz, int ksize) { return wgrad_mask_bytes((long long)n * gx * gy * gz, ksize); }
extern "C" size_t nrpn_conv3d_wgrad_workspace_bytes(int n, int gx, int gy, int gz, int cin, int cout, int wrows, int ksize, int dtype) {
  const long long M = (long long)n * gx * gy * gz;
  const int slices = wgrad_plan(M, wrows, cin, ksize == 3 ? 27 : 1, dtype == NRPN_F32 ? 4 : 2, 0, cout, cin).ksplit;
  return wgrad_mask_bytes(M, ksize) + (size_t)slices * wrows * 4;
}

static int conv3d_wgrad_impl(const void *x, const void *dy, float *gw_packed, float *gbias, long long M, int gx, int gy, int gz,
                             const Segs *segs, int cin, int cout, int wrows, int ksize, int dtype, int accumulate_bias, void *workspace,
                             nrpn_stream_t stream, const unsigned *rows = nullptr, long long total_voxels = 0) {
  // rows != nullptr: row-list form -- M = list length (the K extent), x / dy span `total_voxels` rows and are indexed through the list;
  // the workspace then holds only the bias partials (the tap words ride in the list)
  NRPN_REQUIRE(ksize == 1 || ksize == 3, "conv3d_wgrad: ksize must be 1 or 3 (got %d)", ksize);
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "conv3d_wgrad: bad dtype %d", dtype);
  const int es = dtype == NRPN_F32 ? 4 : 2;
  NRPN_REQUIRE((cin * es) % 16 == 0 && (cout * es) % 16 == 0, "conv3d_wgrad: channel rows must be 16-byte multiples (Cin=%d Cout=%d)", cin, cout);
  NRPN_REQUIRE(x && dy && gw_packed && wrows >= cout, "conv3d_wgrad: bad args");
  WgradArgs a{};
  a.x = x; a.dy = dy; a.gw = gw_packed;
  a.M = M;
  a.X = gx; a.Y = gy; a.Z = gz; a.OX = gx; a.OY = gy; a.OZ = gz;
  if (segs) a.segs = *segs;
  a.Cin = cin; a.Cout = cout; a.wrows = wrows; a.taps = ksize == 3 ? 27 : 1; a.stride = 1; a.kpad = 0;
  hipStream_t st = as_stream(stream);
  NRPN_REQUIRE(workspace, "conv3d_wgrad: needs its workspace (nrpn_conv3d_wgrad_workspace_bytes)");
  const long long span = rows ? total_voxels : a.M;
  NRPN_REQUIRE(span * cin * es < (1ll << 31) && span * cout * es < (1ll << 31) && a.M * 8 < (1ll << 31), "conv3d_wgrad: tensors must stay below 2 GiB");
  a.x_bytes = (unsigned)(span * cin * es); a.dy_bytes = (unsigned)(span * cout * es);
  a.rows = rows;
  a.vmask = reinterpret_cast<const unsigned *>(workspace);
  const bool mask_ready = (accumulate_bias & NRPN_WGRAD_MASK_READY) != 0;     // the caller kept the workspace of an earlier call on this grid
  const bool defer_bias = (accumulate_bias & NRPN_WGRAD_DEFER_BIAS) != 0;     // nrpn_reduce_slices will sum the bias partials
  accumulate_bias &= NRPN_WGRAD_ACC_BIAS;
  if (ksize == 3 && !mask_ready && !rows)
    hipLaunchKernelGGL(tap_mask_kernel, dim3((unsigned)cdiv64(a.M, 256)), dim3(256), 0, st, reinterpret_cast<unsigned *>(workspace), a.M, gx, gy, gz,
                       a.segs);
  float *bias_part = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + (rows ? 0 : wgrad_mask_bytes(a.M, ksize)));
  a.gbias = gbias ? bias_part : nullptr;
  const WgPlan pl = wgrad_plan(a.M, wrows, cin, a.taps, es, 0, cout, cin);
  a.ksplit = pl.ksplit;
  a.slice_stride = (long long)a.taps * wrows * cin;
  int rc = NRPN_OK;
  if (dtype == NRPN_F32) rc = launch_wgrad<float, 0>(a, (cin + 127) / 128, st);
  else if (!pl.big) rc = launch_wgrad<bf16s, 0>(a, (cin + 127) / 128, st);
  else {
    a.ntiles_n = (cin + 255) / 256;
    const int tiles = ((wrows + 255) / 256) * a.ntiles_n * a.taps;
    const size_t lds = 2 * 4 * (size_t)64 * 256;
    if (a.rows) {
      NRPN_LDS(conv_wgrad_big_kernel<true>, (int)lds);
      hipLaunchKernelGGL(conv_wgrad_big_kernel<true>, dim3((unsigned)(tiles * a.ksplit)), dim3(512), lds, st, a);
    } else if (cin == 128 && a.taps == 27) {          // wgrad_plan only says `big` for Cin 128 when the pair form is on
      const int tiles2 = ((wrows + 255) / 256) * 14;
      a.ntiles_n = 1;
      NRPN_LDS((conv_wgrad_big_kernel<false, true>), (int)lds);
      hipLaunchKernelGGL((conv_wgrad_big_kernel<false, true>), dim3((unsigned)(tiles2 * a.ksplit)), dim3(512), lds, st, a);
    } else {
      NRPN_LDS(conv_wgrad_big_kernel<false>, (int)lds);
      hipLaunchKernelGGL(conv_wgrad_big_kernel<false>, dim3((unsigned)(tiles * a.ksplit)), dim3(512), lds, st, a);
    }
    NRPN_LAUNCH_CHECK("conv_wgrad_big");
  }
  if (rc || !gbias || defer_bias) return rc;
  hipLaunchKernelGGL(bias_finalize_kernel, dim3((unsigned)((cout + 255) / 256)), dim3(256), 0, st, bias_part, a.ksplit, wrows, cout, gbias,
                     accumulate_bias);
  NRPN_LAUNCH_CHECK("bias_finalize");
  return NRPN_OK;
}

extern "C" int nrpn_conv3d_wgrad(const void *x, const void *dy, float *gw_packed, float *gbias, int n, int gx, int gy, int gz, int cin,
                                 int cout, int wrows, int ksize, int dtype, int accumulate_bias, void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(n > 0 && gx > 0 && gy > 0 && gz > 0, "conv3d_wgrad: bad sizes");
  return conv3d_wgrad_impl(x, dy, gw_packed, gbias, (long long)n * gx * gy * gz, gx, gy, gz, nullptr, cin, cout, wrows, ksize, dtype,
                           accumulate_bias, workspace, stream);
}

extern "C" int nrpn_conv3d_wgrad_ragged(const void *x, const void *dy, float *gw_packed, float *gbias, int nseg, const int32_t *dims,
                                        int cin, int cout, int wrows, int ksize, int dtype, int accumulate_bias, void *workspace,
                                        nrpn_stream_t stream) {
  Segs sg{};
  long long M = 0;
  if (int rc = fill_segs(sg, nseg, dims, M)) return rc;
  return conv3d_wgrad_impl(x, dy, gw_packed, gbias, M, 1, 1, 1, &sg, cin, cout, wrows, ksize, dtype, accumulate_bias, workspace, stream);
}

// Row-list form of the wgrad: dW[tap] = sum over the `nrows` listed voxels v of dY[v] (x) X[v + off(tap)] (list format: csrc/cone.hip; x / dy
// indexed by voxel id over the ragged space of `dims`).  Slice count / partial layout as nrpn_conv3d_wgrad with n = 1, gx = nrows,
// gy = gz = 1 (nrpn_conv3d_wgrad_slices); workspace >= slices * wrows floats when gbias is requested (bias partials), may be NULL otherwise.
extern "C" int nrpn_conv3d_wgrad_rows(const void *x, const void *dy, float *gw_packed, float *gbias, const uint32_t *rows, int64_t nrows,
                                      int nseg, const int32_t *dims, int cin, int cout, int wrows, int ksize, int dtype, int accumulate_bias,
                                      void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(rows && nrows > 0, "conv3d_wgrad_rows: empty row list (the caller zero-fills the gradient instead)");
  Segs sg{};
  long long total = 0;
  if (int rc = fill_segs(sg, nseg, dims, total)) return rc;
  NRPN_REQUIRE(workspace, "conv3d_wgrad_rows: needs a workspace (>= 256 bytes; slices * wrows floats with a bias gradient)");
  return conv3d_wgrad_impl(x, dy, gw_packed, gbias, nrows, 1, 1, 1, &sg, cin, cout, wrows, ksize, dtype, accumulate_bias | NRPN_WGRAD_MASK_READY,
                           workspace, stream, rows, total);
}

// 1 when a wgrad launch of this shape runs conv_wgrad_big_kernel (256x256 tile), 0 for the 128x128 kernel
extern "C" int nrpn_conv3d_wgrad_plan(int n, int gx, int gy, int gz, int cin, int cout, int wrows, int ksize, int dtype) {
  return wgrad_plan((long long)n * gx * gy * gz, wrows, cin, ksize == 3 ? 27 : 1, dtype == NRPN_F32 ? 4 : 2, 0, cout, cin).big ? 1 : 0;
}

extern "C" int nrpn_conv3d_wgrad_slices(int n, int gx, int gy, int gz, int cin, int cout, int wrows, int ksize, int dtype) {
  return wgrad_plan((long long)n * gx * gy * gz, wrows, cin, ksize == 3 ? 27 : 1, dtype == NRPN_F32 ? 4 : 2, 0, cout, cin).ksplit;
}

static long long stem_out_voxels(int n, int gx, int gy, int gz, int stride) {
  return (long long)n * ((gx - 1) / stride + 1) * ((gy - 1) / stride + 1) * ((gz - 1) / stride + 1);
}

extern "C" int nrpn_stem_wgrad_slices(int n, int gx, int gy, int gz, int cout, int stride, int dtype) {
  if (stem_zrow_ok(gz, cout, stride)) return stem_zrow_slices(stem_out_voxels(n, gx, gy, gz, stride), dtype == NRPN_F32 ? 4 : 2);
  return wgrad_plan(stem_out_voxels(n, gx, gy, gz, stride), cout, nrpn_stem_kpad(dtype), 1, dtype == NRPN_F32 ? 4 : 2, 1, cout, 4).ksplit;
}

// floats of ONE slice partial: z-row form [49][cout][32] (stride 2, even Z, Cout 64), otherwise [cout][Kpad]
extern "C" int64_t nrpn_stem_wgrad_slice_floats(int n, int gx, int gy, int gz, int cout, int stride, int dtype) {
  (void)n; (void)gx; (void)gy;
  return stem_zrow_ok(gz, cout, stride) ? (int64_t)49 * cout * 32 : (int64_t)cout * nrpn_stem_kpad(dtype);
}

extern "C" size_t nrpn_stem_wgrad_workspace_bytes(int n, int gx, int gy, int gz, int cout, int stride, int dtype) {
  return (size_t)nrpn_stem_wgrad_slices(n, gx, gy, gz, cout, stride, dtype) * cout * 4;
}

extern "C" int nrpn_conv3d_stem_wgrad(const void *x, const void *dy, float *gw_packed, float *gbias, int n, int gx, int gy, int gz,
                                      int cout, int stride, int dtype, int accumulate_bias, void *workspace, nrpn_stream_t stream) {
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
  {
    const long long xb = (long long)n * gx * gy * gz * 4 * es, db = a.M * cout * es;
    NRPN_REQUIRE(xb < (1ll << 31) && db < (1ll << 31), "stem wgrad: tensors must stay below 2 GiB");
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)db; a.vmask = nullptr;
  }
  NRPN_REQUIRE(!gbias || workspace, "stem wgrad: the bias gradient needs the workspace (nrpn_stem_wgrad_workspace_bytes)");
  if (stem_zrow_ok(gz, cout, stride)) {
    StemZrArgs z{};
    z.x = x; z.dy = dy; z.gw = gw_packed; z.gbias = gbias ? reinterpret_cast<float *>(workspace) : nullptr;
    z.M = a.M; z.X = gx; z.Y = gy; z.Z = gz; z.OX = a.OX; z.OY = a.OY; z.OZ = a.OZ;
    z.slices = stem_zrow_slices(a.M, es);
    z.x_bytes = a.x_bytes; z.dy_bytes = a.dy_bytes;
    const dim3 grid((unsigned)(7 * z.slices));
    if (dtype == NRPN_F32) {
      constexpr size_t lds_ = 2 * (size_t)(ZrCfg<float>::KV * ZrCfg<float>::RSA + 7 * ZrCfg<float>::KV * ZrCfg<float>::RSB);
      NRPN_LDS((stem_wgrad_zrow_kernel<float>), (int)lds_);
      hipLaunchKernelGGL(stem_wgrad_zrow_kernel<float>, grid, dim3(256), lds_, st, z);
    } else {
      constexpr size_t lds_ = 2 * (size_t)(ZrCfg<bf16s>::KV * ZrCfg<bf16s>::RSA + 7 * ZrCfg<bf16s>::KV * ZrCfg<bf16s>::RSB);
      NRPN_LDS((stem_wgrad_zrow_kernel<bf16s>), (int)lds_);
      hipLaunchKernelGGL(stem_wgrad_zrow_kernel<bf16s>, grid, dim3(256), lds_, st, z);
    }
    NRPN_LAUNCH_CHECK("stem_wgrad_zrow");
    if (!gbias) return NRPN_OK;
    hipLaunchKernelGGL(bias_finalize_kernel, dim3(1), dim3(256), 0, st, z.gbias, z.slices, cout, cout, gbias, accumulate_bias);
    NRPN_LAUNCH_CHECK("bias_finalize");
    return NRPN_OK;
  }
  a.gbias = gbias ? reinterpret_cast<float *>(workspace) : nullptr;
  a.ksplit = wgrad_plan(a.M, cout, a.kpad, 1, es, 1, cout, 4).ksplit;
  a.slice_stride = (long long)cout * a.kpad;
  const int rc = (dtype == NRPN_F32) ? launch_wgrad<float, 1>(a, (a.kpad + 127) / 128, st) : launch_wgrad<bf16s, 1>(a, (a.kpad + 127) / 128, st);
  if (rc || !gbias) return rc;
  hipLaunchKernelGGL(bias_finalize_kernel, dim3((unsigned)((cout + 255) / 256)), dim3(256), 0, st, a.gbias, a.ksplit, cout, cout, gbias,
                     accumulate_bias);
  NRPN_LAUNCH_CHECK("bias_finalize");
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

// packed partial gradients [slices][taps][rows_total][cin] -> reference layout [cout][cin][taps], summed over the slices.
// One block = one output row o and 64 input channels: reads are coalesced along cin, an LDS tile turns (tap, c) into
// (c, tap), and the 64 x taps floats of the reference layout are written contiguously.
__global__ void __launch_bounds__(256) unpack_wgrad_kernel(const float *__restrict__ gp, int cout, int cin, int taps, int rows_total,
                                                           int row_offset, float *__restrict__ gw, int accumulate, int slices,
                                                           long long slice_stride) {
  __shared__ float tile[27][65];
  const int o = blockIdx.y, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < taps * 64; i += 256) {
    const int t = i / 64, c = i % 64;
    float v = 0.f;
    if (c0 + c < cin) {
      const float *src = gp + ((long long)t * rows_total + row_offset + o) * cin + c0 + c;
      for (int s = 0; s < slices; ++s) v += src[s * slice_stride];
    }
    tile[t][c] = v;
  }
  __syncthreads();
  const int ncol = min(64, cin - c0);
  float *dst = gw + ((long long)o * cin + c0) * taps;
  for (int i = threadIdx.x; i < ncol * taps; i += 256) {
    const float v = tile[i % taps][i / taps];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

extern "C" int nrpn_unpack_conv_wgrad(const float *gw_packed, int cout, int cin, int taps, int rows_total, int row_offset, float *gw_ref,
                                      int accumulate, int slices, nrpn_stream_t stream) {
  NRPN_REQUIRE(gw_packed && gw_ref && cout > 0 && cin > 0 && taps > 0 && taps <= 27 && slices > 0 && row_offset + cout <= rows_total,
               "unpack_conv_wgrad: bad args");
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)((cin + 63) / 64), (unsigned)cout), dim3(256), 0, as_stream(stream), gw_packed, cout,
                     cin, taps, rows_total, row_offset, gw_ref, accumulate, slices, (long long)taps * rows_total * cin);
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

__global__ void unpack_stem_wgrad_kernel(const float *__restrict__ gp, int cout, int kpad, float *__restrict__ gw, int accumulate, int slices) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)cout * 4 * 343) return;
  const int tap = (int)(i % 343), c = (int)((i / 343) % 4), o = (int)(i / (343 * 4));
  const float *src = gp + (long long)o * kpad + tap * 4 + c;
  float v = 0.f;
  for (int s = 0; s < slices; ++s) v += src[(long long)s * cout * kpad];
  gw[i] = accumulate ? gw[i] + v : v;
}

extern "C" int nrpn_unpack_stem_wgrad(const float *gw_packed, int cout, int dtype, float *gw_ref, int accumulate, int slices,
                                      int64_t slice_floats, nrpn_stream_t stream) {
  NRPN_REQUIRE(gw_packed && gw_ref && cout > 0 && slices > 0, "unpack_stem_wgrad: bad args");
  if (slice_floats == (int64_t)49 * cout * 32) {     // z-row partials (nrpn_stem_wgrad_slice_floats)
    hipLaunchKernelGGL(unpack_stem_zrow_kernel, dim3((unsigned)cdiv64((long long)cout * 4 * 343, 256)), dim3(256), 0, as_stream(stream), gw_packed,
                       cout, gw_ref, accumulate, slices);
    NRPN_LAUNCH_CHECK("unpack_stem_zrow");
    return NRPN_OK;
  }
  const int kpad = nrpn_stem_kpad(dtype);
  hipLaunchKernelGGL(unpack_stem_wgrad_kernel, dim3((unsigned)cdiv64((long long)cout * 4 * 343, 256)), dim3(256), 0, as_stream(stream),
                     gw_packed, cout, kpad, gw_ref, accumulate, slices);
  NRPN_LAUNCH_CHECK("unpack_stem_wgrad");
  return NRPN_OK;
}

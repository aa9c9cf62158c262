/* Tools-only switches of libnerfrpn_hip.so -- NOT part of the drop-in boundary (include/nerfrpn.h).
 *
 * Process-wide defaults for A/B measurements (tools/, a few kernel-variant tests): atomics, read once per call.  The product path never
 * calls them; a caller that needs a non-default plan passes nrpn_conv_opts to the *_ex entry points of nerfrpn.h, which override these
 * defaults per call.  Kept out of the public header so that "no global state" is what the boundary offers. */
#ifndef NERFRPN_TOOLS_H
#define NERFRPN_TOOLS_H
#include "nerfrpn.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tuning knob: K-step of the k1/k3 implicit-GEMM kernels in bytes per tile row (64 or 128, default 128) */
int nrpn_set_conv_kstep_bytes(int kb);
/* tuning knob: 1 (default) = operands go global -> LDS by LDS-DMA (buffer_load ... lds), 0 = register-staged */
int nrpn_set_conv_lds_dma(int on);
/* tuning knob: tile of the bf16 k1/k3 LDS-DMA kernel -- 0 = per shape, 128 = 128x128 (two workgroups per CU), 256 = wave-specialised
 * 256x128 (4 MFMA + 4 LDS-DMA waves), 512 = 256x256 with 8 waves (default for Cout >= 256 when it yields >= 200 workgroups),
 * 1024 = 256x256 with 4 waves (128x128 per wave) */
int nrpn_set_conv_tile_m(int bm);
/* tools-only default: 1 (default) = the halo form (NRPN_TILE_HALO) is chosen automatically where it applies (bf16 3x3x3, Cout >= 256, grids its
 * 4x8x8 blocks cover with <= 12 % waste and >= 200 workgroups), 0 = only on request */
int nrpn_set_conv_halo_auto(int on);
/* tools-only default of the halo kernel's K order: 1 = taps paired across channel-chunk boundaries whenever Cin % 128 == 0 (108 instead of 112
 * K-steps at Cin = 256), 2 (default) = from Cin 256 up, 0 = 14 K-steps per chunk with a half-empty last one.  nrpn_conv_opts.halo_pairing
 * (1 = on, 2 = off) overrides it per call. */
int nrpn_set_conv_halo_pairing(int mode);
/* tuning knob: 1 (default) = the two waves of a SIMD issue their LDS-DMA in different sub-steps of the 256x256 kernel's K-step */
int nrpn_set_conv_stagger(int on);
/* tuning knob: 1 (default) = mid-size grids (16..199 tiles of 256x256) run the 256x256 kernel on K slices; 0 = 128-row kernel */
int nrpn_set_conv_big_split(int on);
/* tuning knob: 1 (default) = 256x256 wgrad tiles for bf16 layers with Cout, Cin >= 256; 0 = always the 128x128 kernel */
int nrpn_set_wgrad_big_tile(int on);
/* 128x128 wgrad kernel, dense 3x3x3 layers with Cin <= 64: 1 (default) = a workgroup owns a PAIR of taps (B tile = [tap 2t | tap 2t + 1], one shared
 * dY tile, 14 workgroups per (tile, slice)), 0 = one tap per workgroup (the tile a quarter / half full) */
int nrpn_set_wgrad_pack2(int on);
/* gradient sum of squares (nrpn_grad_sumsq), A/B only: form 0 (default) = one contiguous range per block, four 16-byte loads in flight per lane;
 * 1 = the same ranges with eight loads in flight; 2 = grid-stride with eight loads in flight.  grid = blocks launched (64 .. 2048; default 1024).
 * The result is deterministic for a given (form, grid); different settings add the fp32 partials in a different order. */
int nrpn_set_sumsq_form(int form, int grid);
/* bf16 wgrad operand fetch: 1 (default) = ds_read_b64_tr_b16 transpose reads, 0 = scalar 16-bit LDS gathers. */
int nrpn_set_wgrad_transpose_read(int on);
/* row-list conv (nrpn_conv3d_fwd_rows): 1 = lists of >= 96 tiles of 256 rows run the 256x256 tile (measured 0.4 % slower: kept for A/B), 0 (default) = 128-row tiles */
int nrpn_set_rows_big_tile(int on);
/* row-list conv: 1 (default) = the M tiles of a short last round (tiles = k x 512 slots + a remainder <= 256) run on K slices, 0 = whole */
int nrpn_set_rows_tail_split(int on);
/* BatchNorm apply / backward-apply: 1 (default) = the hoisted-parameter kernels with 16-byte accesses, 0 = the general grid-stride kernels (A/B) */
int nrpn_set_bn_fast(int on);
/* BatchNorm channel reductions (statistics / backward sums), bf16: 1 = 8 channels per lane (16-byte loads), 0 (default) = 4 -- measured slower with 8 */
int nrpn_set_bn_reduce_v8(int on);
/* max-pool forward / backward: 1 (default) = multiply-shift index arithmetic + compile-time stride, 0 = the general kernels (A/B, same bits) */
int nrpn_set_pool_fast(int on);
/* GroupNorm apply / backward-apply (bf16, C / groups % 8 == 0): 1 (default) = hoisted-parameter kernels with 16-byte accesses, 0 = general (A/B, same bits) */
int nrpn_set_gn_fast(int on);
/* host evaluation of the kernels' multiply-shift division (csrc/common.h FastDiv): n / d for 0 <= n < 2^31, 1 <= d < 2^31 (-1 outside); tests only */
int64_t nrpn_fastdiv_host(int64_t n, int64_t d);
/* bf16 window attention: 1 (default) = MFMA kernels, 0 = the VALU kernels (always used for fp32) */
int nrpn_set_window_attn_mfma(int on);

#ifdef __cplusplus
}
#endif
#endif

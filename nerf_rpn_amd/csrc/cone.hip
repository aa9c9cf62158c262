// Sampled-anchor cones of the RPN head (training).
//
// The RPN loss reads the head's outputs at the <= rpn_batch_size_per_mesh sampled anchors of a scene only (reference rpn.py:389-420: the
// objectness / regression terms index `sampled_inds` / `sampled_pos_inds`; "During training, boxes pred and scores are unused",
// rpn.py:506).  The head is a chain of D 3x3x3 convolutions + one 1x1x1 output GEMM, so the outputs at a voxel set S0 depend on the
// last hidden map on S0, the one before on S1 = dilate(S0), ..., and the gradients that flow back are non-zero on the same sets.
// These kernels turn the sampled anchor indices into the sorted voxel lists S0 c S1 c ... c SD over the ragged (level, scene) voxel
// space of the head (conv_common.cuh: Segs); the row-list forms of the conv kernels (nrpn_conv3d_fwd_rows / nrpn_conv3d_wgrad_rows) then
// evaluate exactly those rows.  Everything is integer work on ~1e5 voxels: byte maps in L2, ballots, one ordered compaction.
//
// List entry (two u32 per row): { voxel id in the ragged space, in-bounds bits of the 27 taps | segment id << 27 } -- the tap word has
// the format of tap_mask_kernel (conv3d.hip); bit 13 (centre tap) doubles as "row valid".
#include "conv_common.cuh"

namespace {
constexpr int kMaxLevels = 8;
struct ConeGeom {
  Segs segs;                       // (level, scene) segments: seg = level * n_scenes + scene
  long long level_off[kMaxLevels + 1];   // first anchor of each level inside one scene's flat anchor list; [nlevels] = anchors per scene
  int nlevels, n_scenes, A;
};

__global__ void cone_mark_kernel(const long long *__restrict__ pos, const long long *__restrict__ neg, const int *__restrict__ counts,
                                 long long pos_stride, long long neg_stride, ConeGeom g, unsigned char *__restrict__ lvl, int *__restrict__ err) {
  const long long per = pos_stride + neg_stride;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per * g.n_scenes) return;
  const int scene = (int)(i / per);
  const long long j = i - (long long)scene * per;
  long long a;
  if (j < pos_stride) {
    if (j >= counts[3 * scene]) return;
    a = pos[scene * pos_stride + j];
  } else {
    if (j - pos_stride >= counts[3 * scene + 1]) return;
    a = neg[scene * neg_stride + (j - pos_stride)];
  }
  if (a < 0 || a >= g.level_off[g.nlevels]) { atomicOr(err, 1); return; }
  int level = 0;
#pragma unroll
  for (int l = 1; l < kMaxLevels; ++l)
    if (l < g.nlevels && a >= g.level_off[l]) level = l;
  const long long cell = (a - g.level_off[level]) / g.A;
  const int seg = level * g.n_scenes + scene;
  int start = 0, end = 0;
#pragma unroll
  for (int q = 0; q < kMaxSeg; ++q)
    if (q == seg) { start = g.segs.start[q]; end = g.segs.start[q + 1]; }
  if (start + cell >= end) { atomicOr(err, 2); return; }
  lvl[start + cell] = 0;
}

// one dilation step: voxels not yet in a set join S_k when one of their in-bounds 26 neighbours is in S_{k-1}
__global__ void cone_dilate_kernel(unsigned char *__restrict__ lvl, long long total, Segs segs, int k) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= total) return;
  if (lvl[v] != 255) return;
  int x, y, z, X, Y, Z;
  locate_voxel(segs, v, 1, 1, 1, x, y, z, X, Y, Z);
  bool hit = false;
#pragma unroll
  for (int t = 0; t < 27; ++t) {
    const int dx = t / 9 - 1, dy = (t / 3) % 3 - 1, dz = t % 3 - 1;
    if (t == 13) continue;
    if ((unsigned)(x + dx) < (unsigned)X && (unsigned)(y + dy) < (unsigned)Y && (unsigned)(z + dz) < (unsigned)Z) {
      const unsigned char nb = lvl[v + ((long long)dx * Y + dy) * Z + dz];
      hit |= nb < k;            // a neighbour concurrently promoted to k does not count: S_k is a function of S_{k-1} only
    }
  }
  if (hit) lvl[v] = (unsigned char)k;
}

// ordered compaction of S_j = { v : lvl[v] <= j }, j = 0..depth (ascending voxel id), in three small launches: per-workgroup counts, one
// scan of those counts, ordered writes.  (A single-workgroup pass over all voxels took 0.3 ms at 73 k voxels: 72 dependent rounds of
// ballots and barriers on one CU, all of it latency in front of the sampler's read-back.)
__global__ void __launch_bounds__(1024) cone_count_kernel(const unsigned char *__restrict__ lvl, long long total, int depth, int *__restrict__ wg_counts) {
  __shared__ int wave_cnt[kMaxLevels][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long v = (long long)blockIdx.x * 1024 + tid;
  const int l = v < total ? (int)lvl[v] : 255;
#pragma unroll
  for (int j = 0; j < kMaxLevels; ++j) {
    if (j > depth) break;
    const unsigned long long b = __ballot(l <= j);
    if (lane == 0) wave_cnt[j][wave] = __popcll(b);
  }
  __syncthreads();
  if (tid <= depth) {
    int s = 0;
    for (int w = 0; w < 16; ++w) s += wave_cnt[tid][w];
    wg_counts[(long long)blockIdx.x * kMaxLevels + tid] = s;
  }
}

// exclusive scan of the per-workgroup counts (in place) + the totals; one workgroup, thread j owns list j
__global__ void cone_scan_kernel(int *__restrict__ wg_counts, int nwg, int depth, int *__restrict__ counts_out, const int *__restrict__ err) {
  const int j = threadIdx.x;
  if (j <= depth) {
    int run = 0;
    for (int g = 0; g < nwg; ++g) {
      const int c = wg_counts[(long long)g * kMaxLevels + j];
      wg_counts[(long long)g * kMaxLevels + j] = run;
      run += c;
    }
    counts_out[j] = run;
  }
  if (j == 0) counts_out[depth + 1] = *err;      // != 0: a sampled anchor index fell outside the pyramid (caller raises)
}

__global__ void __launch_bounds__(1024) cone_write_kernel(const unsigned char *__restrict__ lvl, long long total, Segs segs, int depth,
                                                          const int *__restrict__ wg_base, unsigned *__restrict__ lists, long long cap) {
  __shared__ int wave_cnt[kMaxLevels][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long v = (long long)blockIdx.x * 1024 + tid;
  const int l = v < total ? (int)lvl[v] : 255;
  unsigned word = 0;
  if (l <= depth) {
    int x, y, z, X, Y, Z;
    const int seg = locate_voxel(segs, v, 1, 1, 1, x, y, z, X, Y, Z);
    word = (unsigned)seg << 27;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
      const int dx = t / 9 - 1, dy = (t / 3) % 3 - 1, dz = t % 3 - 1;
      if ((unsigned)(x + dx) < (unsigned)X && (unsigned)(y + dy) < (unsigned)Y && (unsigned)(z + dz) < (unsigned)Z) word |= 1u << t;
    }
  }
  int my_rank[kMaxLevels];
#pragma unroll
  for (int j = 0; j < kMaxLevels; ++j) {
    if (j > depth) break;
    const unsigned long long b = __ballot(l <= j);
    my_rank[j] = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[j][wave] = __popcll(b);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kMaxLevels; ++j) {
    if (j > depth) break;
    if (l <= j) {
      int off = wg_base[(long long)blockIdx.x * kMaxLevels + j];
      for (int w = 0; w < wave; ++w) off += wave_cnt[j][w];
      const long long slot = off + my_rank[j];
      if (slot < cap) {
        lists[((long long)j * cap + slot) * 2] = (unsigned)v;
        lists[((long long)j * cap + slot) * 2 + 1] = word;
      }
    }
  }
}
}  // namespace

// workspace = [level map: total bytes, padded][error flag][per-workgroup counts: ceil(total / 1024) x kMaxLevels ints]
extern "C" size_t nrpn_cone_workspace_bytes(int64_t total_voxels) {
  return (size_t)((total_voxels + 255) / 256 * 256 + 256 + ((total_voxels + 1023) / 1024) * kMaxLevels * 4);
}

extern "C" int nrpn_cone_build(const int64_t *pos, const int64_t *neg, const int32_t *counts, int n_scenes, int64_t pos_stride,
                               int64_t neg_stride, int nlevels, const int64_t *level_anchor_off, int num_anchors, const int32_t *dims,
                               int depth, uint32_t *lists, int64_t cap, int32_t *counts_out, void *workspace, nrpn_stream_t stream) {
  NRPN_REQUIRE(pos && neg && counts && level_anchor_off && dims && lists && counts_out && workspace, "cone_build: null pointer");
  NRPN_REQUIRE(nlevels >= 1 && nlevels <= kMaxLevels && n_scenes >= 1 && nlevels * n_scenes <= kMaxSeg,
               "cone_build: %d levels x %d scenes do not fit the %d ragged segments of the conv kernels", nlevels, n_scenes, kMaxSeg);
  NRPN_REQUIRE(depth >= 0 && depth < kMaxLevels && num_anchors > 0 && pos_stride >= 0 && neg_stride >= 0 && pos_stride + neg_stride > 0,
               "cone_build: bad depth / anchor count / strides");
  ConeGeom g{};
  long long total = 0;
  if (int rc = fill_segs(g.segs, nlevels * n_scenes, dims, total)) return rc;
  g.nlevels = nlevels; g.n_scenes = n_scenes; g.A = num_anchors;
  for (int l = 0; l <= nlevels; ++l) g.level_off[l] = level_anchor_off[l];
  for (int l = 0; l < nlevels; ++l) {          // the anchors of a level must be its cells x A, every scene of a level the same grid
    const int s0 = l * n_scenes;
    const long long cells = (long long)g.segs.X[s0] * g.segs.Y[s0] * g.segs.Z[s0];
    NRPN_REQUIRE(g.level_off[l + 1] - g.level_off[l] == cells * num_anchors, "cone_build: level %d holds %lld anchors, its grid %lld cells x %d",
                 l, (long long)(g.level_off[l + 1] - g.level_off[l]), cells, num_anchors);
  }
  NRPN_REQUIRE(cap >= total, "cone_build: list capacity %lld below the %lld voxels of the ragged space", (long long)cap, total);
  hipStream_t st = as_stream(stream);
  unsigned char *lvl = reinterpret_cast<unsigned char *>(workspace);
  int *err = reinterpret_cast<int *>(lvl + (total + 255) / 256 * 256);
  NRPN_HIP(hipMemsetAsync(lvl, 0xFF, (size_t)total, st));
  NRPN_HIP(hipMemsetAsync(err, 0, 4, st));
  const long long marks = (pos_stride + neg_stride) * n_scenes;
  hipLaunchKernelGGL(cone_mark_kernel, dim3((unsigned)cdiv64(marks, 256)), dim3(256), 0, st, (const long long *)pos, (const long long *)neg, counts,
                     (long long)pos_stride, (long long)neg_stride, g, lvl, err);
  for (int k = 1; k <= depth; ++k)
    hipLaunchKernelGGL(cone_dilate_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, st, lvl, total, g.segs, k);
  int *wg_counts = err + 64;
  const int nwg = (int)cdiv64(total, 1024);
  hipLaunchKernelGGL(cone_count_kernel, dim3(nwg), dim3(1024), 0, st, lvl, total, depth, wg_counts);
  hipLaunchKernelGGL(cone_scan_kernel, dim3(1), dim3(64), 0, st, wg_counts, nwg, depth, counts_out, (const int *)err);
  hipLaunchKernelGGL(cone_write_kernel, dim3(nwg), dim3(1024), 0, st, lvl, total, g.segs, depth, (const int *)wg_counts, lists, (long long)cap);
  NRPN_LAUNCH_CHECK("cone_build");
  return NRPN_OK;
}

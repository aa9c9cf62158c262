// Rotated 3D RoIAlign (second-stage detector) for gfx950 on channels-last feature maps [N][X][Y][Z][C].
//
// Replaces the reference's CUDA op rotated_roi_3d (nerf_rpn/model/rotated_align/src/cuda_3d/ROIAlignRotated3D_cuda.cu:78-170 forward,
// :235-343 backward; Python binding roi_align_rotate_3d.py:13-58): rois [R,8] = (batch index, cx, cy, cz, w, l, h, theta in degrees),
// output [R][pw][pl][ph][C] = average over an adaptive (or fixed) grid of trilinear samples inside each rotated bin.
//
// Design for the part: the reference runs one thread per (roi, channel, bin) on an NCWLH tensor, so every thread recomputes the
// sample positions and the eight taps of a sample are eight scattered scalars.  Here one thread owns 4 consecutive channels of one
// (roi, bin): the 64 lanes of a wave cover 256 channels of the same bin, every tap is one coalesced 8/16-byte load per lane, and the
// sample geometry is wave-uniform.  The backward scatters through 64-bit FIXED-POINT integer atomics (2^-44 resolution) into a
// workspace and converts once: integer addition is associative, so the gradient is bit-identical from run to run, unlike the
// reference's fp32 atomicAdd (:329-336) -- and needs no RoI sorting.
// Address of feature element (x, y, z) is ((x*Y + y)*Z + z)*C + c for every shape (the reference's index expression is only that
// element on cubic maps -- DESIGN.md, oracle/roialign.c).
#include "common.h"

namespace {
typedef unsigned short bf16s;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned short us4;

template <typename T> struct v4;
template <> struct v4<float> {
  static __device__ __forceinline__ f4 ld(const float *p) { return *reinterpret_cast<const f4 *>(p); }
  static __device__ __forceinline__ void st(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
};
template <> struct v4<bf16s> {
  static __device__ __forceinline__ f4 ld(const bf16s *p) {
    const us4 u = *reinterpret_cast<const us4 *>(p);
    return f4{bf16_bits_to_f32(u[0]), bf16_bits_to_f32(u[1]), bf16_bits_to_f32(u[2]), bf16_bits_to_f32(u[3])};
  }
  static __device__ __forceinline__ void st(bf16s *p, f4 v) {
    *reinterpret_cast<us4 *>(p) = us4{f32_to_bf16_bits(v[0]), f32_to_bf16_bits(v[1]), f32_to_bf16_bits(v[2]), f32_to_bf16_bits(v[3])};
  }
};

struct Roi {
  int batch, gw, gl, gh;
  float cw, cl, ch, bin_w, bin_l, bin_h, start_w, start_l, start_h, cosT, sinT, count;
};

__device__ __forceinline__ Roi roi_setup(const float *r, float scale, int pw, int pl, int ph, int sampling_ratio) {
  Roi o;
  o.batch = (int)r[0];
  o.cw = r[1] * scale; o.cl = r[2] * scale; o.ch = r[3] * scale;
  float rw = r[4] * scale, rl = r[5] * scale, rh = r[6] * scale;
  const float theta = (float)(r[7] * 3.14159265358979323846 / 180.0);
  rw = fmaxf(rw, 1.f); rl = fmaxf(rl, 1.f); rh = fmaxf(rh, 1.f);   // malformed RoIs become 1x1x1 (:108-110)
  o.bin_h = rh / (float)ph; o.bin_l = rl / (float)pl; o.bin_w = rw / (float)pw;
  o.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / ph);
  o.gl = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rl / pl);
  o.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / pw);
  o.start_h = (float)(-rh / 2.0); o.start_l = (float)(-rl / 2.0); o.start_w = (float)(-rw / 2.0);
  o.cosT = cosf(theta); o.sinT = sinf(theta);
  o.count = (float)(o.gh * o.gl * o.gw);
  return o;
}

struct Taps {
  int xl, xh, yl, yh, zl, zh;
  float w[8];
  bool valid;
};

// trilinear_interpolate / _gradient (:13-76, :173-233): drop samples more than one voxel outside, clamp to the border otherwise
__device__ __forceinline__ Taps taps(int width, int length, int height, float x, float y, float z) {
  Taps t;
  t.valid = !(z < -1.0f || z > (float)height || y < -1.0f || y > (float)length || x < -1.0f || x > (float)width);
  if (z <= 0) z = 0;
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  t.zl = (int)z; t.yl = (int)y; t.xl = (int)x;
  if (t.zl >= height - 1) { t.zh = t.zl = height - 1; z = (float)t.zl; } else t.zh = t.zl + 1;
  if (t.yl >= length - 1) { t.yh = t.yl = length - 1; y = (float)t.yl; } else t.yh = t.yl + 1;
  if (t.xl >= width - 1) { t.xh = t.xl = width - 1; x = (float)t.xl; } else t.xh = t.xl + 1;
  const float lz = z - t.zl, ly = y - t.yl, lx = x - t.xl;
  const float hz = 1.f - lz, hy = 1.f - ly, hx = 1.f - lx;
  t.w[0] = lz * hy * hx; t.w[1] = lz * hy * lx; t.w[2] = lz * ly * hx; t.w[3] = lz * ly * lx;
  t.w[4] = hz * hy * hx; t.w[5] = hz * hy * lx; t.w[6] = hz * ly * hx; t.w[7] = hz * ly * lx;
  return t;
}

__device__ __forceinline__ long long tap_voxel(const Taps &t, int k, int length, int height) {
  const int z = (k < 4) ? t.zh : t.zl, y = (k & 2) ? t.yh : t.yl, x = (k & 1) ? t.xh : t.xl;
  return ((long long)x * length + y) * height + z;
}

__device__ __forceinline__ void sample_xyz(const Roi &o, int pw, int pl, int ph, int ix, int iy, int iz, float &x, float &y, float &z) {
  const float zz = o.start_h + ph * o.bin_h + (iz + .5f) * o.bin_h / (float)o.gh;
  const float yy = o.start_l + pl * o.bin_l + (iy + .5f) * o.bin_l / (float)o.gl;
  const float xx = o.start_w + pw * o.bin_w + (ix + .5f) * o.bin_w / (float)o.gw;
  x = xx * o.cosT + yy * o.sinT + o.cw;
  y = yy * o.cosT - xx * o.sinT + o.cl;
  z = zz + o.ch;
}

// 2^44 fixed point in a signed 64-bit accumulator: contributions of 1e-9 (a mean loss over ~1000 RoIs spread over 8..64 samples) keep
// 4+ significant digits, values down to 6e-14 survive, and the sum may reach +-2^19 before it would wrap; a single contribution is
// clamped to +-2^18 so that one outlier cannot wrap the accumulator on its own.
constexpr double kFix = 17592186044416.0;     // 2^44
constexpr float kFixClamp = 262144.f;         // 2^18

// MODE 0: forward  out[bin][c] = mean of samples;  MODE 1: backward scatter of grad_out[bin][c] * w / count (fixed point)
template <typename T, int MODE>
__global__ void __launch_bounds__(64) roi_align_kernel(const T *__restrict__ feat, const float *__restrict__ rois, int width, int length, int height,
                                                       int channels, float scale, int pw_n, int pl_n, int ph_n, int sampling_ratio,
                                                       T *__restrict__ out, const T *__restrict__ grad_out, long long *__restrict__ ws,
                                                       int nbatch) {
  const int bins = pw_n * pl_n * ph_n;
  const int bin = blockIdx.x % bins;
  const long long n = blockIdx.x / bins;
  const int ph = bin % ph_n, pl = (bin / ph_n) % pl_n, pw = bin / (ph_n * pl_n);
  const Roi o = roi_setup(rois + n * 8, scale, pw_n, pl_n, ph_n, sampling_ratio);
  const long long base = (long long)o.batch * width * length * height;
  if (o.batch < 0 || o.batch >= nbatch) {      // a bad batch index would read / atomically write out of bounds: the RoI pools to zeros
    if (MODE == 0) {
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      for (int cg = threadIdx.x * 4; cg < channels; cg += 256) v4<T>::st(out + ((long long)n * bins + bin) * channels + cg, z);
    }
    return;
  }
  for (int cg = threadIdx.x * 4; cg < channels; cg += 256) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 top = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) top = v4<T>::ld(grad_out + ((long long)n * bins + bin) * channels + cg);
    for (int iz = 0; iz < o.gh; ++iz)
      for (int iy = 0; iy < o.gl; ++iy)
        for (int ix = 0; ix < o.gw; ++ix) {
          float x, y, z;
          sample_xyz(o, pw, pl, ph, ix, iy, iz, x, y, z);
          const Taps t = taps(width, length, height, x, y, z);
          if (!t.valid) continue;
          if (MODE == 0) {
            f4 val = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const f4 v = v4<T>::ld(feat + (base + tap_voxel(t, k, length, height)) * channels + cg);
#pragma unroll
              for (int q = 0; q < 4; ++q) val[q] += t.w[k] * v[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += val[q];
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              long long *dst = ws + (base + tap_voxel(t, k, length, height)) * channels + cg;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float g = fminf(fmaxf(top[q] * t.w[k] / o.count, -kFixClamp), kFixClamp);
                if (g != 0.f) atomicAdd(reinterpret_cast<unsigned long long *>(dst + q), (unsigned long long)__double2ll_rn((double)g * kFix));
              }
            }
          }
        }
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = acc[q] / o.count;
      v4<T>::st(out + ((long long)n * bins + bin) * channels + cg, acc);
    }
  }
}

template <typename T>
__global__ void fixed_to_float_kernel(const long long *__restrict__ ws, T *__restrict__ dst, long long count) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x)
    elem<T>::st(dst + i, (float)((double)ws[i] * (1.0 / kFix)));
}

int check(int num_rois, int n, int x, int y, int z, int c, int pw, int pl, int ph, int dtype) {
  NRPN_REQUIRE(num_rois >= 0 && n > 0 && x > 0 && y > 0 && z > 0 && c > 0 && c % 4 == 0 && pw > 0 && pl > 0 && ph > 0,
               "roi_align_rotated_3d: bad sizes (C must be a multiple of 4)");
  NRPN_REQUIRE(dtype == NRPN_F32 || dtype == NRPN_BF16, "roi_align_rotated_3d: bad dtype %d", dtype);
  NRPN_REQUIRE((long long)num_rois * pw * pl * ph < (1ll << 31), "roi_align_rotated_3d: too many bins");
  return 0;
}
}  // namespace

extern "C" int nrpn_roi_align_rotated_3d_fwd(const void *feat, const float *rois, int num_rois, int n, int x, int y, int z, int c,
                                             float spatial_scale, int pw, int pl, int ph, int sampling_ratio, void *out, int dtype,
                                             nrpn_stream_t stream) {
  if (int rc = check(num_rois, n, x, y, z, c, pw, pl, ph, dtype)) return rc;
  if (num_rois == 0) return NRPN_OK;
  NRPN_REQUIRE(feat && rois && out, "roi_align_rotated_3d_fwd: null pointer");
  const dim3 grid((unsigned)(num_rois * pw * pl * ph));
  if (dtype == NRPN_F32)
    hipLaunchKernelGGL((roi_align_kernel<float, 0>), grid, dim3(64), 0, as_stream(stream), (const float *)feat, rois, x, y, z, c, spatial_scale, pw,
                       pl, ph, sampling_ratio, (float *)out, (const float *)nullptr, (long long *)nullptr, n);
  else
    hipLaunchKernelGGL((roi_align_kernel<bf16s, 0>), grid, dim3(64), 0, as_stream(stream), (const bf16s *)feat, rois, x, y, z, c, spatial_scale,
                       pw, pl, ph, sampling_ratio, (bf16s *)out, (const bf16s *)nullptr, (long long *)nullptr, n);
  NRPN_LAUNCH_CHECK("roi_align_rotated_3d_fwd");
  return NRPN_OK;
}

extern "C" size_t nrpn_roi_align_rotated_3d_bwd_workspace_bytes(int n, int x, int y, int z, int c) { return (size_t)n * x * y * z * c * 8; }

extern "C" int nrpn_roi_align_rotated_3d_bwd(const void *grad_out, const float *rois, int num_rois, int n, int x, int y, int z, int c,
                                             float spatial_scale, int pw, int pl, int ph, int sampling_ratio, void *grad_in, void *workspace,
                                             int dtype, nrpn_stream_t stream) {
  if (int rc = check(num_rois, n, x, y, z, c, pw, pl, ph, dtype)) return rc;
  NRPN_REQUIRE(grad_in && workspace && (num_rois == 0 || (grad_out && rois)), "roi_align_rotated_3d_bwd: null pointer");
  hipStream_t st = as_stream(stream);
  const long long count = (long long)n * x * y * z * c;
  NRPN_HIP(hipMemsetAsync(workspace, 0, (size_t)count * 8, st));
  if (num_rois > 0) {
    const dim3 grid((unsigned)(num_rois * pw * pl * ph));
    if (dtype == NRPN_F32)
      hipLaunchKernelGGL((roi_align_kernel<float, 1>), grid, dim3(64), 0, st, (const float *)nullptr, rois, x, y, z, c, spatial_scale, pw, pl, ph,
                         sampling_ratio, (float *)nullptr, (const float *)grad_out, (long long *)workspace, n);
    else
      hipLaunchKernelGGL((roi_align_kernel<bf16s, 1>), grid, dim3(64), 0, st, (const bf16s *)nullptr, rois, x, y, z, c, spatial_scale, pw, pl, ph,
                         sampling_ratio, (bf16s *)nullptr, (const bf16s *)grad_out, (long long *)workspace, n);
  }
  const int blocks = (int)min((long long)8192, (count + 255) / 256);
  if (dtype == NRPN_F32) hipLaunchKernelGGL(fixed_to_float_kernel<float>, dim3(blocks), dim3(256), 0, st, (const long long *)workspace, (float *)grad_in, count);
  else hipLaunchKernelGGL(fixed_to_float_kernel<bf16s>, dim3(blocks), dim3(256), 0, st, (const long long *)workspace, (bf16s *)grad_in, count);
  NRPN_LAUNCH_CHECK("roi_align_rotated_3d_bwd");
  return NRPN_OK;
}

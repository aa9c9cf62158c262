/* ORACLE (test infrastructure only): rotated 3D RoIAlign, forward and backward, as plain C loops.
 *
 * Restates reference nerf_rpn/model/rotated_align/src/cuda_3d/ROIAlignRotated3D_cuda.cu:
 *   trilinear_interpolate (:13-76), RoIAlignRotated3DForward (:78-170), trilinear_interpolate_gradient (:173-233),
 *   RoIAlignRotated3DBackwardFeature (:235-343); layouts of the op: input [N,C,W,L,H], rois [R,8] = (batch index, cx, cy, cz, w, l, h,
 *   theta in DEGREES), output [R,C,pw,pl,ph].
 *
 * PARITY UNPINNED: the CUDA source cannot be built or run here (needs ATen + a CUDA device) and the reference ships no test vectors
 * for it; this restatement is validated by known answers only (tests/test_oracle_golden.py::test_roialign_known_answers).
 *
 * One deliberate divergence, documented in DESIGN.md: the reference addresses the feature map as (x*width + y)*length + z in the
 * forward (:63-70) and with a mix of x*length / x*width in the backward (:329-336), which is the element (x, y, z) of a [W,L,H] map
 * only when W == L == H.  This file (and the HIP kernel) use the layout's own address (x*L + y)*H + z for every shape; on cubic
 * feature maps the two coincide. */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct { int xl, xh, yl, yh, zl, zh; float w[8]; int valid; } taps_t;

/* weights w1..w8 and corner indices of one sample (:173-233); valid = 0 for samples dropped by the out-of-range test (:19-23) */
static taps_t taps(int width, int length, int height, float x, float y, float z) {
  taps_t t;
  memset(&t, 0, sizeof t);
  if (z < -1.0 || z > height || y < -1.0 || y > length || x < -1.0 || x > width) return t;
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
  t.valid = 1;
  return t;
}

/* corner k of the 8 taps -> (x, y, z): order v1..v8 of the reference (:63-70) */
static void corner(const taps_t *t, int k, int *x, int *y, int *z) {
  *z = (k < 4) ? t->zh : t->zl;
  *y = (k & 2) ? t->yh : t->yl;
  *x = (k & 1) ? t->xh : t->xl;
}

typedef struct {
  int batch, gw, gl, gh;
  float cw, cl, ch, bin_w, bin_l, bin_h, start_w, start_l, start_h, cosT, sinT, count;
} roi_t;

static roi_t roi_setup(const float *r, float scale, int pw, int pl, int ph, int sampling_ratio) {
  roi_t o;
  o.batch = (int)r[0];
  o.cw = r[1] * scale; o.cl = r[2] * scale; o.ch = r[3] * scale;
  float rw = r[4] * scale, rl = r[5] * scale, rh = r[6] * scale;
  const float theta = (float)(r[7] * M_PI / 180.0);
  rw = fmaxf(rw, 1.f); rl = fmaxf(rl, 1.f); rh = fmaxf(rh, 1.f);
  o.bin_h = rh / (float)ph; o.bin_l = rl / (float)pl; o.bin_w = rw / (float)pw;
  o.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / ph);
  o.gl = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rl / pl);
  o.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / pw);
  o.start_h = (float)(-rh / 2.0); o.start_l = (float)(-rl / 2.0); o.start_w = (float)(-rw / 2.0);
  o.cosT = cosf(theta); o.sinT = sinf(theta);
  o.count = (float)(o.gh * o.gl * o.gw);
  return o;
}

static void sample_xyz(const roi_t *o, int pw, int pl, int ph, int ix, int iy, int iz, float *x, float *y, float *z) {
  const float zz = o->start_h + ph * o->bin_h + (iz + .5f) * o->bin_h / (float)o->gh;
  const float yy = o->start_l + pl * o->bin_l + (iy + .5f) * o->bin_l / (float)o->gl;
  const float xx = o->start_w + pw * o->bin_w + (ix + .5f) * o->bin_w / (float)o->gw;
  *x = xx * o->cosT + yy * o->sinT + o->cw;
  *y = yy * o->cosT - xx * o->sinT + o->cl;
  *z = zz + o->ch;
}

void oracle_roi_align_rotated_3d_fwd(const float *input, const float *rois, int64_t num_rois, int channels, int width, int length, int height,
                                     float spatial_scale, int pw_n, int pl_n, int ph_n, int sampling_ratio, float *out) {
  for (int64_t n = 0; n < num_rois; ++n) {
    const roi_t o = roi_setup(rois + n * 8, spatial_scale, pw_n, pl_n, ph_n, sampling_ratio);
    for (int c = 0; c < channels; ++c) {
      const float *data = input + ((int64_t)o.batch * channels + c) * height * length * width;
      for (int pw = 0; pw < pw_n; ++pw)
        for (int pl = 0; pl < pl_n; ++pl)
          for (int ph = 0; ph < ph_n; ++ph) {
            float acc = 0.f;
            for (int iz = 0; iz < o.gh; ++iz)
              for (int iy = 0; iy < o.gl; ++iy)
                for (int ix = 0; ix < o.gw; ++ix) {
                  float x, y, z;
                  sample_xyz(&o, pw, pl, ph, ix, iy, iz, &x, &y, &z);
                  const taps_t t = taps(width, length, height, x, y, z);
                  if (!t.valid) continue;
                  float val = 0.f;
                  for (int k = 0; k < 8; ++k) {
                    int cx, cy, cz;
                    corner(&t, k, &cx, &cy, &cz);
                    val += t.w[k] * data[((int64_t)cx * length + cy) * height + cz];
                  }
                  acc += val;
                }
            out[((((int64_t)n * channels + c) * pw_n + pw) * pl_n + pl) * ph_n + ph] = acc / o.count;
          }
    }
  }
}

/* grad_in [N,C,W,L,H] must be zero-filled by the caller */
void oracle_roi_align_rotated_3d_bwd(const float *grad_out, const float *rois, int64_t num_rois, int channels, int width, int length, int height,
                                     float spatial_scale, int pw_n, int pl_n, int ph_n, int sampling_ratio, double *grad_in) {
  for (int64_t n = 0; n < num_rois; ++n) {
    const roi_t o = roi_setup(rois + n * 8, spatial_scale, pw_n, pl_n, ph_n, sampling_ratio);
    for (int c = 0; c < channels; ++c) {
      double *diff = grad_in + ((int64_t)o.batch * channels + c) * height * length * width;
      for (int pw = 0; pw < pw_n; ++pw)
        for (int pl = 0; pl < pl_n; ++pl)
          for (int ph = 0; ph < ph_n; ++ph) {
            const float top = grad_out[((((int64_t)n * channels + c) * pw_n + pw) * pl_n + pl) * ph_n + ph];
            for (int iz = 0; iz < o.gh; ++iz)
              for (int iy = 0; iy < o.gl; ++iy)
                for (int ix = 0; ix < o.gw; ++ix) {
                  float x, y, z;
                  sample_xyz(&o, pw, pl, ph, ix, iy, iz, &x, &y, &z);
                  const taps_t t = taps(width, length, height, x, y, z);
                  if (!t.valid) continue;
                  for (int k = 0; k < 8; ++k) {
                    int cx, cy, cz;
                    corner(&t, k, &cx, &cy, &cz);
                    diff[((int64_t)cx * length + cy) * height + cz] += (double)(top * t.w[k] / o.count);
                  }
                }
          }
    }
  }
}

// Probe: semantics of ds_read_b64_tr_b16 as used by the bf16 wgrad fragment fetch (conv3d.hip wg_frag<bf16s>).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s4v;
constexpr int RS = 320;  // bytes per voxel row (128 bf16 + 64 pad)
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) char tile[32 * RS];
  for (int i = threadIdx.x; i < 32 * 128; i += 64) { int v = i / 128, c = i % 128; *(short*)(tile + v * RS + c * 2) = (short)(v * 1000 + c); }
  __syncthreads();
  const int lane = threadIdx.x, h = lane >> 5, p = lane & 15;
  const int ctile0 = 32, kbase = 16;
  const int cbase = ctile0 + 16 * ((lane >> 4) & 1);
  const int vb = kbase + 8 * h;
  const char* a0 = tile + (vb + (p >> 2)) * RS + (cbase + 4 * (p & 3)) * 2;
  s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a0));
  s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4v __attribute__((address_space(3)))*)(a0 + 4 * RS));
  for (int q = 0; q < 4; ++q) { out[lane * 8 + q] = lo[q]; out[lane * 8 + 4 + q] = hi[q]; }
}
int main() {
  short* d; hipMalloc(&d, 64 * 8 * 2); k<<<1, 64>>>(d); short h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int q = 0; q < 8; ++q) { int want = (16 + 8 * (l >> 5) + q) * 1000 + 32 + (l & 31); if (h[l * 8 + q] != want) ++bad; }
  printf("tr16 probe: %d mismatches of 512\n", bad);
  for (int l = 0; l < 64; l += 5) { printf("lane %2d:", l); for (int q = 0; q < 8; ++q) printf(" %5d", h[l * 8 + q]); printf("\n"); }
  return bad != 0;
}

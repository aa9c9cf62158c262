// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) on gfx950 -- lane -> LDS placement (M0 base + lane*16) and whether
// out-of-range lanes deposit zeros.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* in, unsigned* out, int bytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, bytes, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned off = (unsigned)((wave * 64 + (63 - lane)) * 16);     // reversed source order inside each wave
  if (lane % 5 == 0) off = 0x80000000u;                            // out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, off, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) out[i] = lds[i];
}
int main() {
  unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i;
  unsigned *din, *dout; hipMalloc(&din, 4096); hipMalloc(&dout, 4096); hipMemcpy(din, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 256>>>(din, dout, 4096); unsigned o[1024]; hipMemcpy(o, dout, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    unsigned want = (l % 5 == 0) ? 0u : (unsigned)((w * 64 + (63 - l)) * 4 + e);
    if (o[w * 256 + l * 4 + e] != want) { if (bad < 8) printf("w%d l%d e%d got %08x want %08x\n", w, l, e, o[w*256+l*4+e], want); ++bad; }
  }
  printf("glds probe: %d mismatches of 1024\n", bad);
  return bad != 0;
}

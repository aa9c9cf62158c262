// Register-only MFMA ceiling probe for gfx950: v_mfma_f32_32x32x16_bf16 issued back to back from registers (no LDS, no VMEM inside the
// loop), with the wave-level blocking of the production conv kernels (4 A fragments x 2 B fragments -> 8 accumulators) and two operand
// register sets alternating per sub-step (the production kernels change one register set per sub-step as well).  The operands come from
// global memory once, so the caller chooses what toggles in the MFMA array: zeros, post-ReLU randn, dense randn.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_peak_probe.hip -o tools/libmfma_probe.so   (tools/mfma_peak_probe.py)
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma_loop_kernel(const f4 *__restrict__ ops, float *__restrict__ out, int iters) {
  // 12 fragments of 16 bytes per lane: set 0 = a0..a3, b0, b1; set 1 likewise
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f4 *src = ops + ((size_t)(blockIdx.x * WAVES + wave) % 64) * 12 * 64 + lane;
  f4 fr[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) fr[i] = src[i * 64];
  f16v acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, fr[s * 6 + i]), __builtin_bit_cast(bf8, fr[s * 6 + 4 + j]),
                                                                   acc[i * 2 + j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;      // keeps the loop alive; practically never taken
}

extern "C" int mfma_probe_launch(const void *ops, void *out, int blocks, int waves, int iters, hipStream_t stream) {
  if (waves == 4)
    hipLaunchKernelGGL(mfma_loop_kernel<4>, dim3(blocks), dim3(256), 0, stream, (const f4 *)ops, (float *)out, iters);
  else if (waves == 8)
    hipLaunchKernelGGL(mfma_loop_kernel<8>, dim3(blocks), dim3(512), 0, stream, (const f4 *)ops, (float *)out, iters);
  else
    return 1;
  return (int)hipGetLastError();
}

// Shared host-side plumbing of libnerfrpn_hip.so (error reporting, launch checks, dtype helpers).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/nerfrpn.h"
#include "../../include/nerfrpn_tools.h"

int nrpn_fail(int code, const char *fmt, ...);

#define NRPN_REQUIRE(cond, ...)                                 \
  do {                                                          \
    if (!(cond)) return nrpn_fail(NRPN_ERR_ARG, __VA_ARGS__);   \
  } while (0)

#define NRPN_HIP(call)                                                                          \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return nrpn_fail(NRPN_ERR_LAUNCH, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

#define NRPN_LAUNCH_CHECK(name)                                                                  \
  do {                                                                                           \
    hipError_t e_ = hipGetLastError();                                                           \
    if (e_ != hipSuccess) return nrpn_fail(NRPN_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// Division by a launch-invariant divisor without the division (round 6): the loaders' prologue decomposed every tile row's voxel index with
// three 64-bit divisions and built its 27-bit tap mask with 27 x 3 compares -- ~1 600 dynamic VALU instructions per wave before the first load
// was issued (tools/isa_audit.py: 5 500 static instructions ahead of the first buffer_load), ~3 us of every launch of these kernels and a
// third of the 10^3 / 5^3 / 1x1x1 launches.  For n < 2^31 and 1 <= d < 2^31: with l = ceil(log2 d), m = ceil(2^(31 + l) / d) < 2^32 and
// n / d == (n * m) >> (31 + l) exactly (Granlund & Montgomery, N = 31).  d == 1 has no 32-bit m: the kernel keeps n.
struct FastDiv {
  unsigned m, sh, d;      // q = d == 1 ? n : mulhi(n, m) >> sh
};
static inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f{0u, 0u, d};
  if (d <= 1) return f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                                     // ceil(log2 d), >= 1
  f.m = (unsigned)((((unsigned long long)1 << (31 + l)) + d - 1) / d);
  f.sh = l - 1;                                                    // (n * m) >> (31 + l) = mulhi(n, m) >> (l - 1)
  return f;
}
__device__ __forceinline__ unsigned fastdiv(unsigned n, const FastDiv &f) { return f.d == 1 ? n : (__umulhi(n, f.m) >> f.sh); }


// Raise a kernel's dynamic-LDS limit once per (kernel, device): the attribute belongs to the device the call is made on, so the cache
// is keyed on both (a process-wide "done" flag would leave a second device of the same process at the 64 KB default).  Thread-safe.
int nrpn_ensure_dynamic_lds(const void *kernel, int bytes);
#define NRPN_LDS(kernel, bytes)                                                                         \
  do {                                                                                                  \
    if (int rc_ = nrpn_ensure_dynamic_lds(reinterpret_cast<const void *>(kernel), (int)(bytes))) return rc_; \
  } while (0)

static inline hipStream_t as_stream(nrpn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// bf16 <-> f32 on raw 16-bit storage (round-to-nearest-even), usable in device code.
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// Round-to-nearest-even on the conversion unit (fptrunc -> v_cvt_pk_bf16_f32 on gfx950): one instruction instead of the ~7 VALU
// instructions of the integer formulation (add 0x7fff + lsb, NaN test, shift), on every bf16 output element of every kernel.  Same
// result for every non-NaN input (the kernels keep fp32 denormals: float_denorm_mode_32 = 3); NaNs come out quiet.
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

template <typename T> struct elem;
template <> struct elem<float> {
  static __device__ __forceinline__ float ld(const float *p) { return *p; }
  static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct elem<unsigned short> {
  static __device__ __forceinline__ float ld(const unsigned short *p) { return bf16_bits_to_f32(*p); }
  static __device__ __forceinline__ void st(unsigned short *p, float v) { *p = f32_to_bf16_bits(v); }
};

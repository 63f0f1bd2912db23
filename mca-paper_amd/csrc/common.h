// Shared device helpers for the gfx950 kernels.  Wavefront = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mca_hip.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define WAVE 64

__device__ __forceinline__ float bf2f(u16 b) { return __uint_as_float(((uint32_t)b) << 16); }
// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ u16 f2bf(float f) {
  __bf16 h = (__bf16)f;
  return *reinterpret_cast<u16*>(&h);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, WAVE));
  return v;
}

// transposed LDS read: 16-lane group reads a 4-row x 16-col block of 16-bit elements, lane i gets
// column i of the 4 rows (verified on hardware by tools/probe_layouts.hip)
__device__ __forceinline__ bf16x4 lds_read_tr16(const u16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 __attribute__((address_space(3)))*)(p));
}

// exact (erf) GELU and its derivative (model.py:38 F.gelu default)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

extern int mca_knobs[8];      // A/B measurement knobs, set through mca_debug_set (defined in optim.hip)

static inline hipStream_t as_stream(mca_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int launch_status() { return hipGetLastError() == hipSuccess ? MCA_OK : MCA_E_LAUNCH; }

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
// the same value as ONE v_cvt_pk_bf16_f32 whatever the use of the result (pack2bf relies on the compiler re-fusing two scalar
// conversions, which it does not do when the dword is used as a dword)
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf_pk(float lo, float hi) {
  const bf16x2v h = __builtin_convertvector(f32x2v{lo, hi}, bf16x2v);
  return *reinterpret_cast<const uint32_t*>(&h);
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

// GELU and its derivative from ONE exp and one rcp (fused GEMM epilogues, where erff + expf per element would make the
// epilogue VALU-bound): erf by Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7, whose exp(-z^2) with z = x/sqrt(2) is the
// Gaussian of the derivative as well.
__device__ __forceinline__ void gelu_pair(float x, float& gelu, float& dgelu) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);          // exp(-x^2/2)
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f); p = fmaf(t, p, -0.284496736f); p = fmaf(t, p, 0.254829592f);
  const float erf_abs = fmaf(-p * t, e, 1.f);
  const float cdf = fmaf(copysignf(erf_abs, x), 0.5f, 0.5f);
  gelu = x * cdf;
  dgelu = fmaf(x * 0.3989422804014327f, e, cdf);
}

// ---- XCD-aware bijective remap of a linear workgroup id (guide section 5, "XCD swizzle must be bijective"): workgroup id i
// runs on XCD i % 8; the remap gives every XCD a CONTIGUOUS range of the returned index, so workgroups that share operands
// (neighbouring indices) share an L2
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / 8;
}

extern int mca_knobs[16];      // A/B measurement knobs, set through mca_debug_set (defined in optim.hip)

// Timeline probe of a kernel's debug mode: s_memtime stamps of one wave in a per-file device buffer, read back by tools/
// through mca_dbg_trace_read_<NAME> (exported, but not part of the ABI in include/mca_hip.h).
#define MCA_TRACE_BUFFER(NAME)                                                                                     \
  __device__ unsigned long long mca_trace_##NAME[1024];                                                            \
  extern "C" int mca_dbg_trace_read_##NAME(unsigned long long* dst, int n) {                                       \
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(mca_trace_##NAME), sizeof(unsigned long long) * (n > 1024 ? 1024 : n)) == hipSuccess ? 0 : -1; \
  }

// the "done" flag of a once-per-DEVICE call site: hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute, so a
// process that drives a second GPU must set it there too (one process per GPU is the normal case: flag 0 only)
static inline bool* mca_dev_flag(bool (&flags)[64]) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  return &flags[dev];
}

static inline hipStream_t as_stream(mca_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline int launch_status() { return hipGetLastError() == hipSuccess ? MCA_OK : MCA_E_LAUNCH; }

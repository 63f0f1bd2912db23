// clip_grad_norm_(max_norm) + AdamW over ONE flat fp32 buffer holding every parameter.
// Reference: train_accel_gpu.py:116-118 (accelerator.clip_grad_norm_, torch.optim.AdamW defaults:
// betas (0.9, 0.999), eps 1e-8, weight_decay 0.01 applied to every tensor).  The reference walks 61
// tensors with for-each kernels and syncs the host for the norm; here the norm stays on the device.
#include "common.h"
#include "../../include/mca_hip_debug.h"

// Two launches, no atomics: every block leaves its partial sum in a fixed slot, one block adds the slots in index order.  The
// sum is then the same bits on every launch (an atomicAdd per block made the clip coefficient, and with it the weights of
// data-parallel replicas that hold identical gradients, differ in the last bit from rank to rank).  The slots are words
// 1 .. MCA_SQNORM_WORDS - 1 of the CALLER's buffer (word 0 = the norm): the library keeps no state of its own, so two norms
// on two streams, two engines in one process or a bench A/B cannot meet in a hidden array.
#define SQN_BLOCKS (MCA_SQNORM_WORDS - 2)
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, int64_t n4, int64_t n, float* __restrict__ mca_sqnorm_partials) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += g[i] * g[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) mca_sqnorm_partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void sqnorm_final_kernel(int blocks, float* __restrict__ out, const float* __restrict__ mca_sqnorm_partials) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 64) acc += mca_sqnorm_partials[i];          // lane l: slots l, l + 64, ... in order
  acc = wave_sum(acc);
  if (threadIdx.x == 0) { out[0] = acc; out[MCA_SQNORM_WORDS - 1] = sqrtf(acc); }          // sum g^2 and the norm itself (for logs: no sqrt launch)
}

extern "C" int mca_grad_sqnorm(const float* g, int64_t n, float* sqnorm, mca_stream_t stream) {
  if (!g || !sqnorm || n <= 0) return MCA_E_BADARG;
  if ((uintptr_t)g % 16) return MCA_E_ALIGN;
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256; if (blocks > SQN_BLOCKS) blocks = SQN_BLOCKS; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, n4, n, sqnorm + 1);
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(64), 0, as_stream(stream), (int)blocks, sqnorm, sqnorm + 1);
  return launch_status();
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                     float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                     float wd, float bc1, float bc2, float max_norm, const float* __restrict__ sqnorm,
                                                     const int32_t* __restrict__ skip_flag, const float* __restrict__ hyper) {
  if (skip_flag && *skip_flag != 0) return;          // a non-finite step (mca_nonfinite_flag) never reaches the weights
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }          // per-step values from device memory (graph replay)
  float clip = 1.f;
  if (max_norm > 0.f && sqnorm) {
    const float c = max_norm / (sqrtf(*sqnorm) + 1e-6f);       // torch.nn.utils.clip_grad_norm_
    clip = c < 1.f ? c : 1.f;
  }
  const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * clip;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rbc2 + eps;
    pi -= step * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

extern "C" int mca_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, float bias_corr1, float bias_corr2, float max_norm,
                              const float* sqnorm, const int32_t* skip_flag, const float* hyper, mca_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0) return MCA_E_BADARG;
  int64_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bias_corr1, bias_corr2, max_norm, sqnorm, skip_flag, hyper);
  return launch_status();
}

__global__ void adamw_hyper_kernel(float* __restrict__ hyper, float lr, float bc1, float bc2) {
  if (threadIdx.x == 0) { hyper[0] = lr; hyper[1] = bc1; hyper[2] = bc2; }
}
extern "C" int mca_adamw_hyper(float* hyper, float lr, float bias_corr1, float bias_corr2, mca_stream_t stream) {
  if (!hyper) return MCA_E_BADARG;
  hipLaunchKernelGGL(adamw_hyper_kernel, dim3(1), dim3(64), 0, as_stream(stream), hyper, lr, bias_corr1, bias_corr2);
  return launch_status();
}

int mca_knobs[16] = {0};
extern "C" int mca_debug_set(int key, int value) { if (key >= 0 && key < 16) mca_knobs[key] = value; return 0; }
extern "C" int mca_debug_reset(void) { for (int i = 0; i < 16; i++) mca_knobs[i] = 0; return 0; }

extern "C" const char* mca_version(void) { return "mca_hip 0.1 (gfx950)"; }

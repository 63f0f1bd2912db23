// Ordered hand-off probe (measurement tool, not part of the library): what does it cost to sum fp32 tiles across the key-block
// workgroups of one (sample, head) in a FIXED order, through L2, instead of with atomics or a second pass?  (DESIGN.md section 5,
// "the one-pass backward with an ordered dQ hand-off, priced".)
//
// Model: `chains` chains of `len` workgroups (one workgroup per key block, 256 threads), every chain owns `tiles` tiles of
// 64 x 64 fp32 (16 KB).  Workgroup j of a chain visits the tiles in the order (start_j + k) mod tiles, k = 0 .. tiles - 1,
// start_j = rot * j: rot = 0 is the in-order pipeline (workgroup j trails j - 1 by one tile: fill and drain of len - 1 tiles),
// rot > 0 the rotated walk (every workgroup starts somewhere else; the order of the additions into a tile is still a function of
// the schedule alone).  Per tile a workgroup: (1) spins, BOUNDED, until the workgroup before it in that tile's order has published
// it (agent-scope acquire on a per-(chain, workgroup) progress counter), (2) reads the tile, adds its own 64 x 64 contribution,
// writes it back, (3) publishes (release).  `work` iterations of dependent FMAs per tile stand in for the tile's matrix work.
// MODE 0: the hand-off as described (compiler's agent-scope fences: L2 write-back + invalidate); MODE 3: the same with fences that
// assume producer and consumer share an L2 (one XCD per chain): L1 invalidate + a wait for the stores; MODE 1: no waiting and no tile traffic (the compute alone); MODE 2: tile traffic with fp32
// atomics instead of the ordered read-modify-write (no waiting).
// Every workgroup of a chain has the same blockIdx % 8, i.e. sits on ONE XCD (same_xcd = 1), or the chain is spread over all
// eight (same_xcd = 0).  The grid must not exceed the number of CUs: every workgroup has to be resident for the chain to advance.
//   build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -shared -fPIC -o /tmp/probe_handoff.so tools/probe_handoff.hip
#include <hip/hip_runtime.h>
#include <cstdint>

#define SPIN_LIMIT (1 << 16)          // ~ 0.1 s per wait: a chain that does not advance ends as a wrong result, not as a hang

template <int MODE>
__global__ __launch_bounds__(256) void handoff_kernel(float* tiles_mem, int* progress, int* errors, unsigned long long* cycles, int chains,
                                                      int len, int tiles, int rot, int work, int same_xcd) {
  // blockIdx -> (chain, position): same_xcd: chain c lives on XCD c % 8: its workgroups are blockIdx = (c % 8) + 8 * (slot), slots
  // handed out chain after chain
  const int bid = blockIdx.x;
  int c, j;
  if (same_xcd) {
    const int xcd = bid & 7, slot = bid >> 3;          // `slot`-th workgroup on that XCD
    const int chains_here = (chains - xcd + 7) / 8;     // chains xcd, xcd + 8, ...
    if (slot >= chains_here * len) return;
    c = xcd + 8 * (slot / len); j = slot % len;
  } else {
    if (bid >= chains * len) return;
    c = bid / len; j = bid % len;
  }
  const int tid = threadIdx.x;
  float acc[16];
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 1.0f;          // this workgroup's contribution to every tile: all ones
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long waited = 0;
  for (int k = 0; k < tiles; k++) {
    const int t = (rot * j + k) % tiles;
    // stand-in for the tile's matrix work
    float w = acc[0];
    for (int i = 0; i < work; i++) w = __builtin_fmaf(w, 1.0000001f, 1e-9f);
    asm volatile("" :: "v"(w));          // (keeps the loop)
    float* tp = tiles_mem + ((int64_t)c * tiles + t) * 4096 + tid * 16;
    if (MODE == 0 || MODE == 3) {
      // who adds to tile t before me?  the workgroup p whose k_p = k - d is the largest below k, i.e. the previous one in the
      // order of arrival at t: arrival step of workgroup q at t is (t - rot * q) mod tiles
      int pred = -1, pred_step = -1;
      for (int q = 0; q < len; q++) {
        if (q == j) continue;
        const int step = ((t - rot * q) % tiles + tiles) % tiles;
        if ((step < k || (step == k && q < j)) && (step > pred_step || (step == pred_step && q > pred))) { pred = q; pred_step = step; }
      }
      if (pred >= 0) {
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        int spins = 0;
        if (tid == 0) {
          while (__hip_atomic_load(progress + c * len + pred, MODE == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= pred_step && spins < SPIN_LIMIT) {
            __builtin_amdgcn_s_sleep(2);
            spins++;
          }
          if (spins >= SPIN_LIMIT) atomicAdd(errors, 1);
        }
        __syncthreads();
        if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // buffer_inv sc1: the whole L2's non-local lines
        else asm volatile("buffer_inv sc0" ::: "memory");                          // MODE 3: this CU's L1 only (producer on the same XCD: its data is in the shared L2)
        waited += __builtin_amdgcn_s_memtime() - w0;
      }
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = pred >= 0 ? reinterpret_cast<const f4*>(tp)[e] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; e++) { v[e][0] += acc[4 * e]; v[e][1] += acc[4 * e + 1]; v[e][2] += acc[4 * e + 2]; v[e][3] += acc[4 * e + 3]; }
#pragma unroll
      for (int e = 0; e < 4; e++) reinterpret_cast<f4*>(tp)[e] = v[e];
      if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // buffer_wbl2 sc1: every dirty line of this XCD's L2
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // MODE 3: the write-through L1 has handed the stores to L2
      __syncthreads();
      if (tid == 0) {
        if (MODE == 0) __hip_atomic_store(progress + c * len + j, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(progress + c * len + j, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int e = 0; e < 16; e++) atomicAdd(tp + e, acc[e]);
    }
  }
  if (tid == 0) { cycles[2 * bid] = __builtin_amdgcn_s_memtime() - t0; cycles[2 * bid + 1] = waited; }
}

extern "C" int handoff_run(int mode, float* tiles_mem, int* progress, int* errors, unsigned long long* cycles, int chains, int len, int tiles,
                           int rot, int work, int same_xcd, int grid, float* ms_out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, 0);
  if (mode == 0) hipLaunchKernelGGL(handoff_kernel<0>, dim3(grid), dim3(256), 0, 0, tiles_mem, progress, errors, cycles, chains, len, tiles, rot, work, same_xcd);
  else if (mode == 3) hipLaunchKernelGGL(handoff_kernel<3>, dim3(grid), dim3(256), 0, 0, tiles_mem, progress, errors, cycles, chains, len, tiles, rot, work, same_xcd);
  else if (mode == 1) hipLaunchKernelGGL(handoff_kernel<1>, dim3(grid), dim3(256), 0, 0, tiles_mem, progress, errors, cycles, chains, len, tiles, rot, work, same_xcd);
  else hipLaunchKernelGGL(handoff_kernel<2>, dim3(grid), dim3(256), 0, 0, tiles_mem, progress, errors, cycles, chains, len, tiles, rot, work, same_xcd);
  hipEventRecord(b, 0);
  if (hipEventSynchronize(b) != hipSuccess) return -1;
  hipEventElapsedTime(ms_out, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

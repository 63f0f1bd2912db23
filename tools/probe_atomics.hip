// Atomic-rate probe (measurement tool): how fast does the chip retire fp32 / packed-bf16 / 64-bit integer atomic adds
// with the access pattern of the attention backward's dQ accumulation (a (rows x 512) fp32 matrix, 64-column head slices,
// several workgroups adding to the same tile at different times)?
//   build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -shared -fPIC -o tools/probe_atomics.so tools/probe_atomics.hip
#include <hip/hip_runtime.h>
#include <cstdint>

// grid: (tiles of 64 rows, heads=8, passes).  512 threads: wave w owns the 16x16 block (qb = w&1, db = w>>1) of both
// 32-row sub-tiles, exactly like attn_bwd_kernel.  MODE 0: fp32 atomic per element (lane&15 = column)   1: plain store
// 2: u64 integer atomic, lane owns 4 consecutive columns of one row (2 atomics)   3: pk_bf16 atomic (2 per lane -> 4 columns)
// 4: fp32 atomic, lane owns 4 consecutive columns (4 atomics)  5: f64 atomic add, 2 per lane
template <int MODE>
__global__ __launch_bounds__(512) void atomics_kernel(float* dq, int rows, int ld) {
  const int tile = blockIdx.x, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qb = wave & 1, db = wave >> 1, g4 = lane >> 4;
  const float v = 1.0f + lane * 1e-3f;
  if (MODE == 0 || MODE == 1) {
    float* p = dq + h * 64 + db * 16 + (lane & 15);
#pragma unroll
    for (int sub = 0; sub < 2; sub++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int q = tile * 64 + sub * 32 + qb * 16 + 4 * g4 + e;
        if (q < rows) {
          if (MODE == 0) atomicAdd(p + (int64_t)q * ld, v); else __builtin_nontemporal_store(v, p + (int64_t)q * ld);
        }
      }
  } else if (MODE == 6 || MODE == 7 || MODE == 8) {
    // MODE 6: one instruction = 2 rows x 32 columns (two whole 128-byte lines); wave w owns rows [8w, 8w+8) of the tile,
    // MODE 7: one instruction = 1 row x 64 columns (256 bytes); MODE 8: like 6 with plain stores
    const int W = (MODE == 7) ? 64 : 32;
    const int rsub = lane / W, col = lane % W;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      // 8 instructions x 64 lanes = 512 elements per wave, 64 x 64 tile over 8 waves
      const int idx = e * (64 / W) + rsub;              // 0..(8*64/W - 1): (row, column block) slots of this wave
      const int nblk = 64 / W;                          // column blocks per row
      const int q = tile * 64 + wave * 8 + idx / nblk;
      float* p = dq + (int64_t)q * ld + h * 64 + (idx % nblk) * W + col;
      if (q < rows) { if (MODE == 8) __builtin_nontemporal_store(v, p); else atomicAdd(p, v); }
    }
  } else if (MODE == 9 || MODE == 10) {
    // dense 2-elements-per-lane forms: the wave owns rows [8w, 8w+8) of the 64 x 64 tile, an instruction = 2 rows x 64 columns.
    // MODE 9: u64 integer atomic on an fp32-sized element pair (8 B per lane);  MODE 10: pk_bf16 (4 B per lane, bf16 matrix)
    const int rsub = lane >> 5, cp = lane & 31;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int q = tile * 64 + wave * 8 + 2 * e + rsub;
      if (q >= rows) continue;
      if (MODE == 9) {
        unsigned long long* p64 = reinterpret_cast<unsigned long long*>(dq + (int64_t)q * ld + h * 64 + 2 * cp);
        atomicAdd(p64, (unsigned long long)lane * 0x100000001ull);
      } else {
        typedef __attribute__((ext_vector_type(2))) short s2;
        s2 x; x[0] = 0x3f80; x[1] = 0x3f80;
        uint16_t* p16 = reinterpret_cast<uint16_t*>(dq) + (int64_t)q * ld + h * 64 + 2 * cp;
        asm volatile("global_atomic_pk_add_bf16 %0, %1, off" :: "v"(p16), "v"(x) : "memory");
      }
    }
  } else {
    // transposed block: lane&15 = row inside the 16-row block, g4 = group of 4 columns
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const int q = tile * 64 + sub * 32 + qb * 16 + (lane & 15);
      if (q >= rows) continue;
      float* p = dq + (int64_t)q * ld + h * 64 + db * 16 + 4 * g4;
      if (MODE == 2) {
        unsigned long long* p64 = reinterpret_cast<unsigned long long*>(p);
        atomicAdd(p64, (unsigned long long)lane * 0x100000001ull);
        atomicAdd(p64 + 1, (unsigned long long)lane * 0x100000003ull);
      } else if (MODE == 3) {
        // 4 bf16 columns in 8 bytes (the buffer is viewed as bf16 pairs)
        typedef __attribute__((ext_vector_type(2))) short s2;
        s2 x; x[0] = 0x3f80; x[1] = 0x3f80;
        uint32_t* p32 = reinterpret_cast<uint32_t*>(p);
        asm volatile("global_atomic_pk_add_bf16 %0, %1, off" :: "v"(p32), "v"(x) : "memory");
        asm volatile("global_atomic_pk_add_bf16 %0, %1, off offset:4" :: "v"(p32), "v"(x) : "memory");
      } else if (MODE == 4) {
#pragma unroll
        for (int e = 0; e < 4; e++) atomicAdd(p + e, v);
      } else if (MODE == 5) {
        double* pd = reinterpret_cast<double*>(p);
        atomicAdd(pd, (double)v); atomicAdd(pd + 1, (double)v);
      }
    }
  }
}

extern "C" int atomics_run(int mode, float* dq, int rows, int ld, int passes, float* ms) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid((rows + 63) / 64, 8, passes);
  for (int rep = 0; rep < 2; rep++) {
    if (rep == 1) hipEventRecord(e0, 0);
#define L(M) case M: hipLaunchKernelGGL(atomics_kernel<M>, grid, dim3(512), 0, 0, dq, rows, ld); break;
    switch (mode) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) default: return -1; }
    if (rep == 0) hipDeviceSynchronize();
  }
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  hipEventElapsedTime(ms, e0, e1);
  return (int)hipGetLastError();
}

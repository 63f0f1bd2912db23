// bf16 MFMA GEMMs for every Linear of the MCA step (forward, data-gradient and weight-gradient).
//
//   mca_gemm_nt     C[M,N]  = A[M,K] · B[N,K]^T      both operands K-contiguous (Linear forward with
//                                                     x[M,K], W[N,K]; data-gradient with the transposed
//                                                     bf16 weight copy)
//   mca_gemm_nt_lnres / _geglu_fwd / _geglu_bwd       the same GEMM with a fused epilogue (residual LayerNorm
//                                                     recomputed; FF1 + GEGLU; W2 data gradient + GEGLU')
//   mca_gemm_tn_acc C[N,K] += A[R,N]^T · B[R,K]      both operands reduction-major (weight gradient:
//                                                     dW = dY^T · X), operands fetched from LDS with the
//                                                     gfx950 transposed read ds_read_b64_tr_b16
//
// Kernels (all v_mfma_f32_32x32x16_bf16, operands global -> LDS by global_load_lds_dwordx4 with the XOR swizzle
// applied to the source address and again on the fragment reads, counted s_waitcnt vmcnt so the DMA stays in flight
// across the raw s_barrier of a k-step):
//   gemm_nt_persist_kernel   256x128 tiles walked by one workgroup per CU; next tile's DMA under a wave-private,
//                            all-asm epilogue; bf16 / plain fp32 outputs and the two GEGLU fusions (M >= 2048)
//   gemm_nt_256_kernel       256x128x64, three stages, one tile per workgroup; fp32 + residual outputs (residual tile
//                            prefetched into registers; optional LayerNorm recompute), fallback for odd shapes
//   gemm_nt_glds_kernel      128x128x64, two stages: small M
//   gemm_tn_256x256_kernel / gemm_tn_256_kernel / gemm_tn_kernel   weight gradient, split over rows + fp32 atomics
// Workgroup ids are remapped (xcd_remap) so that workgroups sharing operands sit on one XCD / L2.
#include "common.h"

#define g_knob mca_knobs     // A/B measurement knobs (mca_debug_set, optim.hip)

#define BM 128
#define BN 128
#define BK 64

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

// Persistent kernel: tile index -> (row tile tm, column tile tn).  Row tiles are taken in groups of PS_PANELS: index =
// (group, tn, tm in group), so the 32 workgroups of an XCD (contiguous indices, xcd_remap) work on PS_PANELS activation
// panels x 8 weight tiles at a time (1 MiB + 1 MiB of the 4 MiB L2) and a panel group stays resident while the column
// tiles sweep past it; in row-major tile order all 22 column tiles of FF1 (2.9 MiB of weights) are live at once and the
// activation panels are re-fetched 9 times (PMC FETCH_SIZE x 2: 756 MB per launch against 86 MB of operands; 436 MB
// with the groups; step -0.36 ms in-process; 4, 8 and 16 panels per group measure the same, 2 is worse).
#define PS_PANELS 4
__device__ __forceinline__ void ps_tile_decode(int tile, int tiles_m, int tiles_n, int flat, int& tm, int& tn) {
  if (flat == 1) { tm = tile / tiles_n; tn = tile % tiles_n; return; }
  const int P = flat > 1 ? flat : PS_PANELS;
  const int per_group = P * tiles_n;
  const int g = tile / per_group;
  const int gh = (g + 1) * P <= tiles_m ? P : tiles_m - g * P;          // panels in this (last) group
  const int rem = tile - g * per_group;
  tn = rem / gh; tm = g * P + rem % gh;
}

// LDS image of a [128 rows][64 bf16] tile (128-byte rows): 16-byte chunk c of row r lives at chunk
// c ^ ((r >> 1) & 7): a 16-lane group of ds_read_b128 (16 distinct rows, same logical chunk) then covers
// all 64 banks exactly once.
__device__ __forceinline__ int nt_off(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 7)) << 3); }

// Fused GEGLU backward (model.py:35-38 autograd) for 8 consecutive columns n..n+7 of row m of dg = dY·W2 (fp32 in v):
// h = [a | gate] (bf16, row stride ldh, halves ip apart);  dh_a = dg*gelu(gate),  dh_gate = dg*a*gelu'(gate).
__device__ __forceinline__ void geglu_bwd_piece(const float (&v)[8], const u16* __restrict__ h, u16* __restrict__ dh, int64_t ldh,
                                                int ip, int64_t m, int n) {
  const bf16x8 av = *reinterpret_cast<const bf16x8*>(h + m * ldh + n);
  const bf16x8 gv = *reinterpret_cast<const bf16x8*>(h + m * ldh + ip + n);
  bf16x8 da, dgt;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const float g = bf2f((u16)gv[e]), a = bf2f((u16)av[e]);
    float ge, dge;
    gelu_pair(g, ge, dge);
    da[e] = (short)f2bf(v[e] * ge);
    dgt[e] = (short)f2bf(v[e] * a * dge);
  }
  *reinterpret_cast<bf16x8*>(dh + m * ldh + n) = da;
  *reinterpret_cast<bf16x8*>(dh + m * ldh + ip + n) = dgt;
}

// Epilogue through LDS (BK = 64 kernels: the 64 KiB of operand buffers hold exactly one 128x128 fp32 tile).  The
// accumulators are written lane = column / register = row, then every thread reads whole 16-byte pieces of rows,
// adds bias / residual with 16-byte loads and stores 16 bytes: 4x (fp32) or 8x (bf16) fewer store instructions,
// every global access a full 128-byte-line segment of a row.
// Second half of the LDS epilogue: the fp32 C tile (ROWS x 128, row-major in cs) leaves as whole 16-byte row pieces, bias /
// residual / GEGLU-backward fused; THREADS threads, every global access a full 128-byte-line segment of a row.
template <bool OUT_BF16, int RES, int EPI, int ROWS, int THREADS>
__device__ __forceinline__ void nt_epilogue_rows(const float* __restrict__ cs, void* __restrict__ Cv, int64_t ldc,
                                                 const float* __restrict__ bias, const float* __restrict__ residual,
                                                 int64_t ldres, int64_t res_period, int M, int N, int m0, int n0, int tid) {
  constexpr int EPT = OUT_BF16 ? 8 : 4;          // elements per thread per piece
  constexpr int PPR = 128 / EPT;                 // pieces per row
#pragma unroll 4
  for (int it = 0; it < (ROWS * PPR) / THREADS; it++) {
    const int id = tid + THREADS * it;
    const int row = id / PPR, c0 = (id % PPR) * EPT;
    const int m = m0 + row, n = n0 + c0;
    if (m >= M || n >= N) continue;
    float v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(cs + row * 128 + c0 + e);
      v[e] = t[0]; v[e + 1] = t[1]; v[e + 2] = t[2]; v[e + 3] = t[3];
    }
    const bool full = n + EPT <= N;
    if (EPI == 1) {          // GEGLU backward: Cv = dh, residual = h (bf16), ldres = row stride of both, N = ip (multiple of 8)
      if (OUT_BF16) geglu_bwd_piece(reinterpret_cast<const float(&)[8]>(v), reinterpret_cast<const u16*>(residual),
                                    reinterpret_cast<u16*>(Cv), ldres, N, (int64_t)m, n);
      continue;
    }
    if (bias) {
#pragma unroll
      for (int e = 0; e < EPT; e++) if (full || n + e < N) v[e] += bias[n + e];
    }
    if (RES != 0) {
      const int64_t rr = RES == 2 ? (int64_t)(m % (int)res_period) : (int64_t)m;
      const float* rp = residual + rr * ldres + n;
      if (full && (((uintptr_t)rp) & 15) == 0) {
#pragma unroll
        for (int e = 0; e < EPT; e += 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(rp + e);
          v[e] += t[0]; v[e + 1] += t[1]; v[e + 2] += t[2]; v[e + 3] += t[3];
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; e++) if (n + e < N) v[e] += rp[e];
      }
    }
    if (OUT_BF16) {
      u16* cp = reinterpret_cast<u16*>(Cv) + (int64_t)m * ldc + n;
      if (full && (((uintptr_t)cp) & 15) == 0) {
        uint4 pk;
        pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); pk.z = pack2bf(v[4 % EPT], v[5 % EPT]); pk.w = pack2bf(v[6 % EPT], v[7 % EPT]);
        *reinterpret_cast<uint4*>(cp) = pk;
      } else {
#pragma unroll
        for (int e = 0; e < EPT; e++) if (n + e < N) cp[e] = f2bf(v[e]);
      }
    } else {
      float* cp = reinterpret_cast<float*>(Cv) + (int64_t)m * ldc + n;
      if (full && (((uintptr_t)cp) & 15) == 0) {
        *reinterpret_cast<f32x4*>(cp) = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int e = 0; e < EPT; e++) if (n + e < N) cp[e] = v[e];
      }
    }
  }
}

template <bool OUT_BF16, int RES, int EPI>
__device__ __forceinline__ void nt_epilogue_lds(f32x16 (&acc)[2][2], float* __restrict__ cs, void* __restrict__ Cv, int64_t ldc,
                                                const float* __restrict__ bias, const float* __restrict__ residual,
                                                int64_t ldres, int64_t res_period, int M, int N, int m0, int n0, int wm, int wn,
                                                int l31, int lh, int tid) {
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        cs[row * 128 + wn * 64 + j * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();
  nt_epilogue_rows<OUT_BF16, RES, EPI, 128, 256>(cs, Cv, ldc, bias, residual, ldres, res_period, M, N, m0, n0, tid);
}

// ---------------------------------------------------------------------------------------------------------
// NT kernel with direct global->LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
// The LDS destination of one wave-instruction is linear (wave base + lane*16 B), so the XOR swizzle is applied
// to the per-lane SOURCE address and again on the fragment reads (cdna guide, rule 21).
// ---------------------------------------------------------------------------------------------------------
template <int BKT> __device__ __forceinline__ int gl_sw(int r) { return BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
template <int BKT> __device__ __forceinline__ int gl_off(int r, int c) { return r * BKT + ((c ^ gl_sw<BKT>(r)) << 3); }

// One BK = 64 step of a wavefront's 64x64 sub-tile: 4 k16-steps of 2x2 MFMA 32x32x16.  The fragment reads are
// inline asm with hand-counted lgkmcnt so that the reads of step ks+1 are in flight while the MFMAs of step ks
// issue (hipcc otherwise re-uses one register set and waits lgkmcnt(0) in front of every group of 4 MFMAs).
// Byte address of (row r, k16-step ks): P ^ (32*ks) with P = tile_base + r*128 + 16*(lh ^ swizzle(r)); the XOR
// form holds because every base is a multiple of 128 bytes.
struct NtFragAddr { unsigned a[2], b[2]; };
__device__ __forceinline__ NtFragAddr nt_frag_addr(const u16* lds_base, int a_elem_off, int b_elem_off, int wm, int wn, int l31, int lh) {
  NtFragAddr f;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)lds_base;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int ra = wm * 64 + i * 32 + l31, rb = wn * 64 + i * 32 + l31;
    f.a[i] = base + 2u * (unsigned)a_elem_off + (unsigned)(ra * 128) + 16u * (unsigned)(lh ^ ((ra >> 1) & 7));
    f.b[i] = base + 2u * (unsigned)b_elem_off + (unsigned)(rb * 128) + 16u * (unsigned)(lh ^ ((rb >> 1) & 7));
  }
  return f;
}
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#define NT_DSREAD(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
template <bool SWAP = false>          // SWAP: C^T = B.A^T (lane = row of C, registers = 4 consecutive columns)
__device__ __forceinline__ void nt_compute_step(const NtFragAddr& f, unsigned stage_byte_off, f32x16 (&acc)[2][2]) {
  u32x4v fa[2][2], fb[2][2];          // [parity][i]
  const unsigned a0 = f.a[0] + stage_byte_off, a1 = f.a[1] + stage_byte_off;
  const unsigned b0 = f.b[0] + stage_byte_off, b1 = f.b[1] + stage_byte_off;
#define NT_ISSUE(KS, PAR)                                                                            \
  NT_DSREAD(fa[PAR][0], a0 ^ (32u * (KS))); NT_DSREAD(fb[PAR][0], b0 ^ (32u * (KS)));                \
  NT_DSREAD(fa[PAR][1], a1 ^ (32u * (KS))); NT_DSREAD(fb[PAR][1], b1 ^ (32u * (KS)));
#define NT_MFMA(PAR)                                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; i++)                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; j++)                                                    \
      acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fb[PAR][j]), \
                                                                 *reinterpret_cast<const bf16x8*>(&fa[PAR][i]), acc[i][j], 0, 0, 0) \
                       : __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fa[PAR][i]), \
                                                                 *reinterpret_cast<const bf16x8*>(&fb[PAR][j]), acc[i][j], 0, 0, 0);
  NT_ISSUE(0, 0)
  NT_ISSUE(1, 1)
  asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
  NT_MFMA(0)
  __builtin_amdgcn_sched_barrier(0);
  NT_ISSUE(2, 0)
  asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
  NT_MFMA(1)
  __builtin_amdgcn_sched_barrier(0);
  NT_ISSUE(3, 1)
  asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
  NT_MFMA(0)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
  NT_MFMA(1)
#undef NT_ISSUE
#undef NT_MFMA
}

template <bool OUT_BF16, int RES, int BKT, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_glds_kernel(
    const u16* __restrict__ A, int64_t lda, const u16* __restrict__ B, int64_t ldb, void* __restrict__ Cv,
    int64_t ldc, const float* __restrict__ bias, const float* __restrict__ residual, int64_t ldres,
    int64_t res_period, int M, int N, int K, int tiles_n, int nwg) {
  constexpr int CH = BKT / 8;                 // 16-byte chunks per tile row
  constexpr int TILE = 128 * BKT;             // elements per operand tile
  constexpr int NI = CH / 2;                  // wave-instructions per operand tile per wave
  __shared__ __attribute__((aligned(16))) u16 lds[4 * TILE];
  u16* As = lds;
  u16* Bs = lds + 2 * TILE;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const u16* ga[NI];
  const u16* gb[NI];
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int p = (i * 4 + wave) * 64 + lane;           // linear chunk position inside the tile image
    const int r = p / CH, cp = p % CH;
    const int c = cp ^ gl_sw<BKT>(r);                   // logical chunk stored at this position
    int ra = m0 + r; if (ra > M - 1) ra = M - 1;
    int rb = n0 + r; if (rb > N - 1) rb = N - 1;
    ga[i] = A + (int64_t)ra * lda + c * 8;
    gb[i] = B + (int64_t)rb * ldb + c * 8;
  }
  auto stage = [&](int k0, int buf) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      u16* da = As + buf * TILE + (i * 4 + wave) * 512;
      u16* db = Bs + buf * TILE + (i * 4 + wave) * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + k0),
                                       (__attribute__((address_space(3))) void*)da, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + k0),
                                       (__attribute__((address_space(3))) void*)db, 16, 0, 0);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int nkt = K / BKT;
  const NtFragAddr frag = nt_frag_addr(lds, 0, 2 * TILE, wm, wn, l31, lh);
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage((kt + 1) * BKT, cur ^ 1);
    nt_compute_step(frag, (unsigned)cur * (unsigned)(TILE * 2), acc);
    __syncthreads();          // waits for this wave's LDS-DMA (vmcnt(0)) and for every wave's reads of `cur`
  }
  static_assert(BKT == 64, "the LDS epilogue needs the 64 KiB of operand buffers");
  nt_epilogue_lds<OUT_BF16, RES, EPI>(acc, reinterpret_cast<float*>(lds), Cv, ldc, bias, residual, ldres, res_period, M, N, m0, n0,
                                 wm, wn, l31, lh, tid);
}

// ---------------------------------------------------------------------------------------------------------
// NT kernel, large-M variant: 256x128x64 tile, 8 wavefronts (4x2, 64x64 each), THREE LDS stages (144 KiB) filled
// by global_load_lds two k-steps ahead; one raw s_barrier per k-step with a counted s_waitcnt vmcnt so the DMA of
// the next stage stays in flight across the barrier (cdna guide: "Pipelining across barriers").  25 % fewer
// global->LDS bytes per flop than the 128x128 tile.
// ---------------------------------------------------------------------------------------------------------
#define BM2 256
__device__ __forceinline__ void wait_vmcnt(int n) {          // s_waitcnt needs an immediate
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
  }
}
// PF = 1 (fp32 output with a full-row residual, whole column tile inside N): the residual tile (256x128 fp32 =
// 16 float4 per thread) is fetched into registers two pieces per k-step during the first 8 k-steps, so the epilogue
// does not start with a 128 KiB read.  vmcnt counts DMA and residual loads together, in issue order: the wait in
// front of step kt lets everything issued after stage kt's DMA stay in flight.
// EPI = 2 (with PF = 1): the residual is LayerNorm(x) RECOMPUTED from the layer's saved pre-norm tensor x (= `residual`)
// and its row statistics: (x - ln_mean[m]) * ln_rstd[m] * ln_gamma[n] — the normed fp32 tensor is then never written
// by the LayerNorm kernel nor read here (same bytes read, 166 MB less written per LayerNorm at b = 32).  The 256 rows'
// statistics wait in 2 KiB of LDS behind the three stages.
template <bool OUT_BF16, int RES, int PF, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_256_kernel(
    const u16* __restrict__ A, int64_t lda, const u16* __restrict__ B, int64_t ldb, void* __restrict__ Cv,
    int64_t ldc, const float* __restrict__ bias, const float* __restrict__ residual, int64_t ldres,
    int64_t res_period, int M, int N, int K, int tiles_n, int nwg, const float* __restrict__ ln_mean = nullptr,
    const float* __restrict__ ln_rstd = nullptr, const float* __restrict__ ln_gamma = nullptr) {
  extern __shared__ __attribute__((aligned(16))) u16 lds2[];
  constexpr int STAGE = (BM2 + BN) * 64;          // elements per stage: A tile then B tile
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int m0 = tm * BM2, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const u16* ga[4];
  const u16* gb[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p = (i * 8 + wave) * 64 + lane, r = p >> 3, c = (p & 7) ^ gl_sw<64>(r);
    int ra = m0 + r; if (ra > M - 1) ra = M - 1;
    ga[i] = A + (int64_t)ra * lda + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = (i * 8 + wave) * 64 + lane, r = p >> 3, c = (p & 7) ^ gl_sw<64>(r);
    int rb = n0 + r; if (rb > N - 1) rb = N - 1;
    gb[i] = B + (int64_t)rb * ldb + c * 8;
  }
  auto stage = [&](int k0, int st) {
    u16* base = lds2 + st * STAGE;
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + (i * 8 + wave) * 512), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + BM2 * 64 + (i * 8 + wave) * 512), 16, 0, 0);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int nkt = K / 64;
  const NtFragAddr frag = nt_frag_addr(lds2, 0, BM2 * 64, wm, wn, l31, lh);
  f32x4 rres[PF ? 16 : 1];
  auto res_issue = [&](int it) -> f32x4 {          // piece `it` of this thread's share of the residual tile
    const int id = tid + 512 * it, row = id >> 5, c0 = (id & 31) * 4;
    // rows past M are CLAMPED, not skipped: the counted vmcnt waits below assume that every wave issues every one of
    // these loads (a tile crossing M used to skip them and then waited for too few of its DMA loads: a rare stale-LDS
    // race in the last 64 rows of the out-proj / FF2 / data-gradient GEMMs, caught by the bit-exactness test)
    int m = m0 + row; if (m > M - 1) m = M - 1;
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(residual + (int64_t)m * ldres + n0 + c0));          // streamed once
  };
  float* lnstat = reinterpret_cast<float*>(lds2 + 3 * STAGE);          // EPI == 2: [256] mean, [256] rstd
  float ln_val = 0.f;
  if (EPI == 2) {          // issued before the DMA so that its (compiler-counted) wait leaves the DMA in flight
    int m = m0 + (tid & 255); if (m > M - 1) m = M - 1;
    ln_val = tid < 256 ? ln_mean[m] : ln_rstd[m];
  }
  stage(0, 0);
  if (nkt > 1) stage(64, 1);
  if (EPI == 2) lnstat[tid] = ln_val;          // visible to every wave after the k-loop's barriers
  int st = 0;
#define NT256_STEP(KT, WAITN)                                                                  \
  {                                                                                            \
    wait_vmcnt(WAITN);                                                                         \
    __builtin_amdgcn_s_barrier();                                                              \
    if ((KT) + 2 < nkt) { int s2 = st + 2; if (s2 >= 3) s2 -= 3; stage(((KT) + 2) * 64, s2); } \
    nt_compute_step(frag, (unsigned)st * (unsigned)(STAGE * 2), acc);                          \
    st = st == 2 ? 0 : st + 1;                                                                 \
  }
  if (PF) {
    // nkt >= 8 is guaranteed by the host for PF kernels
    NT256_STEP(0, 6)   rres[0] = res_issue(0);   rres[1 % (PF ? 16 : 1)] = res_issue(1);
    NT256_STEP(1, 8)   rres[2 % (PF ? 16 : 1)] = res_issue(2);   rres[3 % (PF ? 16 : 1)] = res_issue(3);
    NT256_STEP(2, 10)  rres[4 % (PF ? 16 : 1)] = res_issue(4);   rres[5 % (PF ? 16 : 1)] = res_issue(5);
    NT256_STEP(3, 10)  rres[6 % (PF ? 16 : 1)] = res_issue(6);   rres[7 % (PF ? 16 : 1)] = res_issue(7);
    NT256_STEP(4, 10)  rres[8 % (PF ? 16 : 1)] = res_issue(8);   rres[9 % (PF ? 16 : 1)] = res_issue(9);
    NT256_STEP(5, 10)  rres[10 % (PF ? 16 : 1)] = res_issue(10); rres[11 % (PF ? 16 : 1)] = res_issue(11);
    NT256_STEP(6, 10)  rres[12 % (PF ? 16 : 1)] = res_issue(12); rres[13 % (PF ? 16 : 1)] = res_issue(13);
    NT256_STEP(7, nkt > 8 ? 10 : 4) rres[14 % (PF ? 16 : 1)] = res_issue(14); rres[15 % (PF ? 16 : 1)] = res_issue(15);
    for (int kt = 8; kt < nkt; kt++) {
      const int more = kt + 1 < nkt ? 6 : 0;
      NT256_STEP(kt, kt == 8 ? 4 + more : (kt == 9 ? 2 + more : more))
    }
  } else {
    for (int kt = 0; kt < nkt; kt++) {
      // this wave's DMA for stage kt has landed once at most the 6 loads of stage kt+1 are outstanding
      NT256_STEP(kt, kt + 1 < nkt ? 6 : 0)
    }
  }
#undef NT256_STEP
  __syncthreads();
  // epilogue through LDS: 256x128 fp32 = 128 KiB
  float* cs = reinterpret_cast<float*>(lds2);
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        cs[row * 128 + wn * 64 + j * 32 + l31] = acc[i][j][r];
      }
  __syncthreads();
  if (PF) {
    // fast epilogue: residual already in registers, whole 128-column tile valid, 16-byte stores
    float lnstat_r[PF ? 16 : 1][2];
    f32x4 lng = f32x4{1.f, 1.f, 1.f, 1.f};
    if (EPI == 2) {
      lng = *reinterpret_cast<const f32x4*>(ln_gamma + n0 + (tid & 31) * 4);
#pragma unroll
      for (int it = 0; it < 16; it++) {
        const int row = (tid + 512 * it) >> 5;
        lnstat_r[it % (PF ? 16 : 1)][0] = lnstat[row]; lnstat_r[it % (PF ? 16 : 1)][1] = lnstat[256 + row];
      }
    }
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int id = tid + 512 * it, row = id >> 5, c0 = (id & 31) * 4;
      const int m = m0 + row;
      if (m < M) {
        f32x4 v = *reinterpret_cast<const f32x4*>(cs + row * 128 + c0);
        f32x4 r = rres[it % (PF ? 16 : 1)];
        if (EPI == 2) {
          const float mu = lnstat_r[it % (PF ? 16 : 1)][0], rs = lnstat_r[it % (PF ? 16 : 1)][1];
#pragma unroll
          for (int e = 0; e < 4; e++) r[e] = (r[e] - mu) * rs * lng[e];
        }
        v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
        if (bias) { v[0] += bias[n0 + c0]; v[1] += bias[n0 + c0 + 1]; v[2] += bias[n0 + c0 + 2]; v[3] += bias[n0 + c0 + 3]; }
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Cv) + (int64_t)m * ldc + n0 + c0));
      }
    }
    return;
  }
  nt_epilogue_rows<OUT_BF16, RES, EPI, BM2, 512>(cs, Cv, ldc, bias, residual, ldres, res_period, M, N, m0, n0, tid);
}
#define NT256_LDS_BYTES (3 * (BM2 + BN) * 64 * 2)
#define NT256LN_LDS_BYTES (NT256_LDS_BYTES + 2048)

// ---------------------------------------------------------------------------------------------------------
// PERSISTENT NT kernel (large M, N % 128 == 0): one 512-thread workgroup per CU walks 256x128 tiles.
// Why: with K = 512 a tile is only 8 k-steps; in the one-tile-per-workgroup kernels every CU loads, computes and
// stores in step with all the others, so HBM idles while the matrix pipes run and vice versa, and each tile pays
// its first-load latency and the drain of its stores (a k-loop-less build of the QKV GEMM still took 108 of its 224 us).  Here
//   * the global->LDS DMA of the NEXT tile's first two k-steps is issued before the epilogue of the current one
//     (into stage slots 0 and 1),
//   * the epilogue is wave-private (no workgroup barrier): each wave moves its 64x64 block through an 8 KiB scratch
//     area (the free third stage slot + 16 KiB: 160 KiB of LDS in all) and leaves as whole 128-byte row segments;
//     its global stores drain while the next tile's k-loop runs.
// The MFMA operands are swapped (C^T = B.A^T) so that a lane holds 4 CONSECUTIVE columns of one row: the
// accumulators go to the scratch area as 8/16-byte pieces instead of single elements.
// All epilogue memory instructions are inline asm: (1) hipcc would put s_waitcnt vmcnt(0) in front of its own LDS
// reads while an LDS-DMA is in flight, (2) the counted vmcnt waits of the next k-loop need the exact number of
// younger stores (CDNA4: loads, stores and LDS-DMA retire in issue order).
// vmcnt bookkeeping per wave and tile, in issue order: ... [DMA of the last k-step: 6] [NLD residual / h / bias loads]
// (k-loop goes on: the last two steps wait vmcnt(6 + NLD) and vmcnt(NLD)) [DMA next tile k-step 0: 6] [DMA k-step 1: 6]
// [NST stores]; k-step 0 of the next tile waits vmcnt(6 + NST), k-step 1 vmcnt(NST + 6) (the DMA of k-step 2 is issued
// in between), later steps vmcnt(6).  Tiles that cross M (predicated stores: unknown count) drain with vmcnt(0).
// ---------------------------------------------------------------------------------------------------------
// The epilogue traffic (C tiles, the h / dh tiles of the GEGLU forms: 0.5-0.9 GB per launch) is streamed with the
// non-temporal hint so that it does not evict the activation panels and weight tiles the k-loops re-read from L2
// (GEGLU backward fetched 778 MB against 540 MB of operands; step -0.28 ms in an alternating-process A/B).
#define PS_NT " nt"
#define PS_GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" PS_NT : "=&v"(dst) : "v"(ptr) : "memory")
// s_nop: a VMEM store of more than 64 bits needs a wait state before a VALU may overwrite its data VGPRs; hipcc
// inserts it for its own stores but cannot see into inline asm
#define PS_GSTORE(ptr, val) asm volatile("global_store_dwordx4 %0, %1, off" PS_NT "\n\ts_nop 1" ::"v"(ptr), "v"(val) : "memory")
#define PS_DSW128(addr, val) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(val) : "memory")
#define PS_DSW64(addr, val) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(val) : "memory")
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
template <int N> __device__ __forceinline__ void ps_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#define NTPS_LDS_BYTES (3 * (BM2 + BN) * 64 * 2 + 16384)

// timeline probe (TRACE build only: mca-paper_amd/build.py --trace; knob 0 = 8): workgroup 0 / wave 0 writes s_memtime stamps,
// read back by tools/trace_persist.py; the product build carries neither the stamps nor the clock probe of the grouped kernel
MCA_TRACE_BUFFER(gemm)
#ifdef MCA_TRACE_BUILD
#define PS_STAMP() do { if (tracing && ti < 1024) mca_trace_gemm[ti++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PS_STAMP() do { } while (0)
#endif

// MODE 0: bf16 out;  1: fp32 out;  2: fp32 out + full-row fp32 residual;  3: fused GEGLU backward (bf16 out, see
// mca_gemm_nt_geglu_bwd: C = dh, residual = h, ldres = row stride of both, N = ip);  4: fused GEGLU forward (see
// mca_gemm_nt_geglu_fwd: B = W1 [2*ip, K], a column tile = 64 "a" rows + the 64 "gate" rows of the same columns,
// C = h [M, 2*ip] bf16, residual = g [M, ip] bf16 with row stride ldres, N = ip, tiles_n = ip / 64)
template <int MODE, bool BIAS>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(
    const u16* __restrict__ A, int64_t lda, const u16* __restrict__ B, int64_t ldb, void* __restrict__ Cv,
    int64_t ldc, const float* __restrict__ bias, const float* __restrict__ residual, int64_t ldres,
    int M, int N, int K, int tiles_n, int nwg, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 lds2[];
  constexpr int STAGE = (BM2 + BN) * 64;          // elements per stage: A tile then B tile
  constexpr int NST = MODE == 0 ? 8 : (MODE == 4 ? 12 : 16);         // global stores per wave and tile
  constexpr int NPRE = (MODE == 2 || MODE == 3) ? 16 : 1;
  constexpr int NLD = ((MODE == 2 || MODE == 3) ? 16 : 0) + (BIAS ? (MODE == 0 ? 2 : 1) : 0);          // epilogue input loads per wave and tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nkt = K / 64;
  const NtFragAddr frag = nt_frag_addr(lds2, 0, BM2 * 64, wm, wn, l31, lh);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)lds2;
  const unsigned scratch = lds_base + 2u * (unsigned)(STAGE * 2) + (unsigned)wave * 8192u;   // stage slot 2 and the 16 KiB after it

  const u16* ga[4];
  const u16* gb[2];
  const int tiles_m = nwg / tiles_n;
  const int flat_order = (dbg >> 5) & 31;          // knob 0 = 32: row-major tile order; 32*P: P panels per group (A/B)
  auto tile_ptrs = [&](int tile) {
    int tm, tn;
    ps_tile_decode(tile, tiles_m, tiles_n, flat_order, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int p = (i * 8 + wave) * 64 + lane, r = p >> 3, c = (p & 7) ^ gl_sw<64>(r);
      int ra = m0 + r; if (ra > M - 1) ra = M - 1;
      ga[i] = A + (int64_t)ra * lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int p = (i * 8 + wave) * 64 + lane, r = p >> 3, c = (p & 7) ^ gl_sw<64>(r);
      // N % 128 == 0 (MODE 4: N % 64 == 0): no clamp.  MODE 4: tile rows 0..63 = "a" rows, 64..127 = "gate" rows
      const int rb = MODE == 4 ? (r < 64 ? tn * 64 + r : N + tn * 64 + (r - 64)) : n0 + r;
      gb[i] = B + (int64_t)rb * ldb + c * 8;
    }
  };
  auto stage = [&](int k0, int st) {
    u16* base = lds2 + st * STAGE;
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + (i * 8 + wave) * 512), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + BM2 * 64 + (i * 8 + wave) * 512), 16, 0, 0);
  };

  int v = blockIdx.x;                              // virtual block id: blockIdx.x + it * gridDim.x (same XCD every time)
  if (v >= nwg) return;
  const bool tracing = (dbg & 8) && blockIdx.x == 0 && tid == 0;
  int ti = 0;
  tile_ptrs(xcd_remap(v, nwg));
  stage(0, 0);
  stage(64, 1);
  bool first = true;
  while (v < nwg) {
    const int tile = xcd_remap(v, nwg);
    int tm, tn;
    ps_tile_decode(tile, tiles_m, tiles_n, flat_order, tm, tn);
    const int m0 = tm * BM2, n0 = tn * BN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // ---------------- k-loop: three-stage ring, DMA two k-steps ahead (as gemm_nt_256_kernel) ----------------
    // this wave's 64x64 block (MODE 4: columns of h: "a" part for wn = 0, "gate" part for wn = 1)
    const int mw = m0 + wm * 64, nw = MODE == 4 ? (wn == 0 ? tn * 64 : N + tn * 64) : n0 + wn * 64;
    const bool edge = m0 + BM2 > M;
    u32x4v pre[NPRE];                              // residual (fp32) / h (bf16 a | gate) pieces of this lane
    u32x4v bvec[2];
    // The epilogue's inputs (NLD loads per wave) are requested three k-steps before the end of the k-loop, right after the
    // last DMA of this tile: they land under the last two k-steps.
    auto preload = [&]() {
      if (BIAS) {
        // MODE 0: this lane's pieces cover columns nw + 8*(lane & 7) .. +7; fp32 modes: nw + 4*(lane & 15) .. +3
        const float* bp = bias + nw + (MODE == 0 ? 8 * (lane & 7) : 4 * (lane & 15));
        PS_GLOAD(bvec[0], bp);
        if (MODE == 0) PS_GLOAD(bvec[1], bp + 4);
      }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int u = 0; u < 8; u++) {
            int m = mw + i * 32 + (lane >> 4) + 4 * u; if (m > M - 1) m = M - 1;
            PS_GLOAD(pre[(i * 8 + u) % NPRE], residual + (int64_t)m * ldres + nw + 4 * (lane & 15));
          }
      }
      if (MODE == 3) {
        const u16* h = reinterpret_cast<const u16*>(residual);
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int u = 0; u < 4; u++) {
            int m = mw + i * 32 + (lane >> 3) + 8 * u; if (m > M - 1) m = M - 1;
            const u16* hp = h + (int64_t)m * ldres + nw + 8 * (lane & 7);
            PS_GLOAD(pre[((i * 4 + u) * 2) % NPRE], hp);                  // a
            PS_GLOAD(pre[((i * 4 + u) * 2 + 1) % NPRE], hp + N);          // gate
          }
      }
    };
    int st = 0;
    PS_STAMP();
    for (int kt = 0; kt < nkt; kt++) {
      // stage kt has landed once only the operations issued after its DMA are outstanding (host guarantees nkt >= 5)
      if (kt < 2) { if (first) ps_wait_vm<6>(); else ps_wait_vm<6 + NST>(); }
      else if (kt == nkt - 2) ps_wait_vm<6 + NLD>();          // DMA of the last stage + the epilogue's loads
      else if (kt == nkt - 1) ps_wait_vm<NLD>();
      else ps_wait_vm<6>();
      PS_STAMP();
      __builtin_amdgcn_s_barrier();
      PS_STAMP();
      if (kt + 2 < nkt) { int s2 = st + 2; if (s2 >= 3) s2 -= 3; stage((kt + 2) * 64, s2); }
      if (NLD > 0 && kt == nkt - 3) preload();
      nt_compute_step<true>(frag, (unsigned)st * (unsigned)(STAGE * 2), acc);
      st = st == 2 ? 0 : st + 1;
      PS_STAMP();
    }
    __builtin_amdgcn_s_barrier();                  // every wave has read its last fragments: all three slots are free
    // The fp32 epilogues hand the accumulators straight to inline-asm ds_write_b128: hipcc does not know the asm reads
    // MFMA results and inserts no wait states (it does for its own v_cvt_pk in the bf16 modes).  The last four MFMAs were
    // issued just before the barrier and take 4 x 32 cycles; without a next tile nothing else sits in between (a
    // one-tile-per-workgroup launch returned stale accumulators once in a few runs).
    if (MODE != 0 && MODE != 4)
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    PS_STAMP();
    // ---------------- epilogue: acc[i][j][4q + e] = C[mw + 32i + l31][nw + 32j + 8q + 4lh + e] ----------------
    // next tile's first two k-steps: in flight during the whole epilogue
    const int vn = v + gridDim.x;
    const bool has_next = vn < nwg;
    if (has_next) {
      tile_ptrs(xcd_remap(vn, nwg));
      stage(0, 0);
      stage(64, 1);
    }
    if (NLD > 0) { if (has_next) ps_wait_vm<12>(); else ps_wait_vm<0>(); }          // the epilogue's loads have landed (requested 3 k-steps ago)
    PS_STAMP();
    if (MODE == 0 || MODE == 4) {
      // bf16 payload: 64 rows x 128 B; 16-byte chunk c of row r at chunk c ^ (r & 7)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            u32x2v pk;
            pk[0] = pack2bf(acc[i][j][4 * q], acc[i][j][4 * q + 1]); pk[1] = pack2bf(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            const int row = i * 32 + l31, ch = j * 4 + q;
            PS_DSW64(scratch + (unsigned)(row * 128 + ((ch ^ (row & 7)) << 4) + 8 * lh), pk);
          }
      u32x4v o[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int row = (lane >> 3) + 8 * u, ch = lane & 7;
        NT_DSREAD(o[u], scratch + (unsigned)(row * 128 + ((ch ^ (row & 7)) << 4)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int m = mw + (lane >> 3) + 8 * u;
        if (BIAS) {          // the accumulators were rounded before the bias: add it and round again (bias GEMMs write fp32 in this model)
          float f[8];
#pragma unroll
          for (int e = 0; e < 4; e++) { f[2 * e] = __uint_as_float(o[u][e] << 16); f[2 * e + 1] = __uint_as_float(o[u][e] & 0xffff0000u); }
#pragma unroll
          for (int e = 0; e < 8; e++) f[e] += __uint_as_float(bvec[e >> 2][e & 3]);
#pragma unroll
          for (int e = 0; e < 4; e++) o[u][e] = pack2bf(f[2 * e], f[2 * e + 1]);
        }
        u16* cp = reinterpret_cast<u16*>(Cv) + (int64_t)m * ldc + nw + 8 * (lane & 7);
        if (m < M) PS_GSTORE(cp, o[u]);
      }
      if (MODE == 4) {
        // g = a * gelu(gate): the partner wave (same rows, other column half) holds the other operand: both blocks are
        // in the scratch areas now; wave wn takes rows 32*wn .. 32*wn + 31 of the pair's 64 rows
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned sa = lds_base + 2u * (unsigned)(STAGE * 2) + (unsigned)(wave & ~1) * 8192u, sg = sa + 8192u;
        u32x4v av[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int row = 32 * wn + (lane >> 3) + 8 * u, ch = lane & 7;
          NT_DSREAD(av[u], sa + (unsigned)(row * 128 + ((ch ^ (row & 7)) << 4)));
          NT_DSREAD(gv[u], sg + (unsigned)(row * 128 + ((ch ^ (row & 7)) << 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        u16* gout = const_cast<u16*>(reinterpret_cast<const u16*>(residual));
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int m = m0 + wm * 64 + 32 * wn + (lane >> 3) + 8 * u;
          u32x4v gg;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float a0 = __uint_as_float(av[u][e] << 16), a1 = __uint_as_float(av[u][e] & 0xffff0000u);
            const float g0 = __uint_as_float(gv[u][e] << 16), g1 = __uint_as_float(gv[u][e] & 0xffff0000u);
            float ge0, ge1, unused;
            gelu_pair(g0, ge0, unused); gelu_pair(g1, ge1, unused);
            gg[e] = pack2bf(a0 * ge0, a1 * ge1);
          }
          u16* gp = gout + (int64_t)m * ldres + tn * 64 + 8 * (lane & 7);
          if (m < M) PS_GSTORE(gp, gg);
        }
      }
    } else {
      // fp32 payload, one pass per row block i: 32 rows x 256 B; 16-byte chunk c of row r at chunk c ^ (r & 15)
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            u32x4v w;
#pragma unroll
            for (int e = 0; e < 4; e++) w[e] = __float_as_uint(acc[i][j][4 * q + e]);
            const int ch = j * 8 + 2 * q + lh;
            PS_DSW128(scratch + (unsigned)(l31 * 256 + ((ch ^ (l31 & 15)) << 4)), w);
          }
        if (MODE != 3) {
          u32x4v o[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int row = (lane >> 4) + 4 * u, ch = lane & 15;
            NT_DSREAD(o[u], scratch + (unsigned)(row * 256 + ((ch ^ (row & 15)) << 4)));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int m = mw + i * 32 + (lane >> 4) + 4 * u;
            u32x4v ov;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              float t = __uint_as_float(o[u][e]);
              if (BIAS) t += __uint_as_float(bvec[0][e]);
              if (MODE == 2) t += __uint_as_float(pre[(i * 8 + u) % NPRE][e]);
              ov[e] = __float_as_uint(t);
            }
            float* cp = reinterpret_cast<float*>(Cv) + (int64_t)m * ldc + nw + 4 * (lane & 15);
            if (m < M) PS_GSTORE(cp, ov);
          }
        } else {
          // GEGLU backward: pieces of 8 columns (two adjacent 16-byte chunks of dg), h = [a | gate]
          u32x4v o[4][2];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int row = (lane >> 3) + 8 * u, ch = 2 * (lane & 7);
            NT_DSREAD(o[u][0], scratch + (unsigned)(row * 256 + ((ch ^ (row & 15)) << 4)));
            NT_DSREAD(o[u][1], scratch + (unsigned)(row * 256 + (((ch + 1) ^ (row & 15)) << 4)));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int m = mw + i * 32 + (lane >> 3) + 8 * u;
            const u32x4v av = pre[((i * 4 + u) * 2) % NPRE], gv = pre[((i * 4 + u) * 2 + 1) % NPRE];
            u32x4v da, dgt;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const float d0 = __uint_as_float(o[u][e >> 1][2 * (e & 1)]), d1 = __uint_as_float(o[u][e >> 1][2 * (e & 1) + 1]);
              const float a0 = __uint_as_float(av[e] << 16), a1 = __uint_as_float(av[e] & 0xffff0000u);
              const float g0 = __uint_as_float(gv[e] << 16), g1 = __uint_as_float(gv[e] & 0xffff0000u);
              float ge0, dge0, ge1, dge1;
              gelu_pair(g0, ge0, dge0); gelu_pair(g1, ge1, dge1);
              da[e] = pack2bf(d0 * ge0, d1 * ge1);
              dgt[e] = pack2bf(d0 * a0 * dge0, d1 * a1 * dge1);
            }
            u16* cp = reinterpret_cast<u16*>(Cv) + (int64_t)m * ldc + nw + 8 * (lane & 7);
            if (m < M) { PS_GSTORE(cp, da); PS_GSTORE(cp + N, dgt); }
          }
        }
      }
    }
    if (edge) ps_wait_vm<0>();
    PS_STAMP();
    first = false;
    v = vn;
  }
}


// ---------------------------------------------------------------------------------------------------------
// PERSISTENT NT kernel with 256 x 256 tiles (bf16 output, N % 256 == 0): the k-loop of these GEMMs is bound by the
// L2 -> LDS path per CU, and a 256x256 tile moves 2/3 of the operand bytes per flop of a 256x128 one.  8 wavefronts as
// 4 (m) x 2 (n), each 64 x 128 = 2 x 4 accumulators (128 VGPRs, operands swapped: lane = row); k-steps of 32 in FIVE
// slots of 32 KiB (DMA three to four steps ahead, software-pipelined fragment reads: see the k-loop); the next tile's first
// three stages are issued before the epilogue, which uses slots 3 and 4 as 8 KiB of scratch per wave (two passes of 32 rows x
// 256 B).  vmcnt per wave: 4 loads per stage, 16 stores per tile.
// ---------------------------------------------------------------------------------------------------------
#define P2_STAGE (512 * 32)          // elements per stage: A image [256][32] then B image [256][32]
#define P2_LDS_BYTES (5 * P2_STAGE * 2)
// GEGLU = true: fused FF1 + GEGLU forward (see mca_gemm_nt_geglu_fwd): B = W1 [2*N, K] with N = ip, a column tile = 128 "a"
// rows + the 128 "gate" rows of the same columns (wave column wn = 0 holds a, wn = 1 gate), C = h [M, 2*N], G = g [M, N].
template <bool GEGLU>
__global__ __launch_bounds__(512) void gemm_nt_persist256_kernel(
    const u16* __restrict__ A, int64_t lda, const u16* __restrict__ B, int64_t ldb, u16* __restrict__ C, int64_t ldc,
    u16* __restrict__ G, int64_t ldg, int M, int N, int K, int tiles_n, int nwg, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 lds2[];
  constexpr int NST = GEGLU ? 24 : 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nkt = K / 32;
  const int tiles_m = nwg / tiles_n;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)lds2;
  const unsigned scratch = lds_base + 3u * (unsigned)(P2_STAGE * 2) + (unsigned)wave * 8192u;          // slot 3 and the 32 KiB after it

  // fragment byte addresses inside a stage: row R (64-byte rows), k16-step ks: R*64 + ((lh ^ sw(R)) << 4) ^ (32*ks)
  unsigned fa_addr[2], fb_addr[4];
#pragma unroll
  for (int i = 0; i < 2; i++) { const int r = wm * 64 + i * 32 + l31; fa_addr[i] = lds_base + (unsigned)(r * 64 + ((lh ^ gl_sw<32>(r)) << 4)); }
#pragma unroll
  for (int j = 0; j < 4; j++) { const int r = wn * 128 + j * 32 + l31; fb_addr[j] = lds_base + 256u * 64u + (unsigned)(r * 64 + ((lh ^ gl_sw<32>(r)) << 4)); }

  const u16* ga[2];
  const u16* gb[2];
  auto tile_ptrs = [&](int tile) {
    int tm, tn;
    ps_tile_decode(tile, tiles_m, tiles_n, 0, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int p = (i * 8 + wave) * 64 + lane, r = p >> 2, c = (p & 3) ^ gl_sw<32>(r);
      int ra = m0 + r; if (ra > M - 1) ra = M - 1;
      ga[i] = A + (int64_t)ra * lda + c * 8;
      // N % 256 == 0 (GEGLU: N % 128 == 0): no clamp.  GEGLU: tile rows 0..127 = "a" rows, 128..255 = "gate" rows
      const int rb = GEGLU ? (r < 128 ? tn * 128 + r : N + tn * 128 + (r - 128)) : n0 + r;
      gb[i] = B + (int64_t)rb * ldb + c * 8;
    }
  };
  auto stage = [&](int k0, int st) {
    u16* base = lds2 + st * P2_STAGE;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + (i * 8 + wave) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb[i] + k0),
                                       (__attribute__((address_space(3))) void*)(base + 256 * 32 + (i * 8 + wave) * 512), 16, 0, 0);
    }
  };

  int v = blockIdx.x;
  if (v >= nwg) return;
  tile_ptrs(xcd_remap(v, nwg));
  stage(0, 0); stage(32, 1); stage(64, 2);
  bool first = true;
  while (v < nwg) {
    const int tile = xcd_remap(v, nwg);
    int tm, tn;
    ps_tile_decode(tile, tiles_m, tiles_n, 0, tm, tn);
    const int m0 = tm * 256, n0 = tn * 256;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    // FIVE slots of 32 KiB (the fifth is the half of the epilogue scratch that a k-loop does not need): stage kt lives in slot
    // kt % 5, stages 0..2 were issued by the previous tile's epilogue (or the prologue), stage 3 goes out at the top of the tile and
    // stage kt + 4 in the middle of step kt.  The step is software-pipelined as in tn_256x256_tile: the six fragment reads of a
    // k16-step sit in the MFMA gaps of the k16-step before it, and the step's one barrier between its two MFMA groups.
    // vmcnt, in issue order per wave: S0 S1 S2 [NST epilogue stores of the previous tile] S3 S4 ...: a stage has landed once only
    // the operations issued after its DMA are outstanding (host: nkt >= 6).
    u32x4v fa[2][2], fb[2][4];          // [k16-step][block]
#define P2_SB __builtin_amdgcn_sched_barrier(0);
#define P2_RA(KS, I, SO) NT_DSREAD(fa[KS][I], (fa_addr[I] + (SO)) ^ (32u * (KS))); P2_SB
#define P2_RB(KS, J, SO) NT_DSREAD(fb[KS][J], (fb_addr[J] + (SO)) ^ (32u * (KS))); P2_SB
#define P2_MM(KS, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&fb[KS][J]), \
                                                                            *reinterpret_cast<const bf16x8*>(&fa[KS][I]), acc[I][J], 0, 0, 0); P2_SB
#define P2_GROUP_READING(KS, KN, SO, MID)                                                                                    \
      P2_MM(KS, 0, 0) MID P2_RA(KN, 0, SO) P2_MM(KS, 0, 1) P2_RB(KN, 0, SO) P2_MM(KS, 0, 2) P2_RB(KN, 1, SO) P2_MM(KS, 0, 3)   \
      P2_RB(KN, 2, SO) P2_MM(KS, 1, 0) P2_RB(KN, 3, SO) P2_MM(KS, 1, 1) P2_RA(KN, 1, SO) P2_MM(KS, 1, 2) P2_MM(KS, 1, 3)
    if (first) ps_wait_vm<8>(); else ps_wait_vm<8 + NST>();          // S0 has landed
    __builtin_amdgcn_s_barrier();                                     // ... for everyone, and every wave has left the previous epilogue
    stage(96, 3);
    P2_SB
    P2_RA(0, 0, 0u) P2_RB(0, 0, 0u) P2_RB(0, 1, 0u) P2_RB(0, 2, 0u) P2_RB(0, 3, 0u) P2_RA(0, 1, 0u)
    int slot = 0, prev = 4;
    for (int kt = 0; kt + 1 < nkt; kt++) {
      const unsigned so = (unsigned)slot * (unsigned)(P2_STAGE * 2);
      const int nxt = slot == 4 ? 0 : slot + 1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); P2_SB
      P2_GROUP_READING(0, 1, so, )
      // stage kt + 1 has landed once only what was issued after it is outstanding; past the barrier every wave has finished with
      // stage kt - 1, whose slot takes stage kt + 4
      if (kt < 2) { if (first) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ps_wait_vm<8 + NST>(); } }
      else if (kt + 3 < nkt) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      P2_SB
      const unsigned son = (unsigned)nxt * (unsigned)(P2_STAGE * 2);
      const bool fetch = kt + 4 < nkt;
      P2_GROUP_READING(1, 0, son, if (fetch) stage((kt + 4) * 32, prev); P2_SB)
      prev = slot; slot = nxt;
    }
    {          // the last stage: nothing left to wait for or to fetch
      const unsigned so = (unsigned)slot * (unsigned)(P2_STAGE * 2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); P2_SB
      P2_GROUP_READING(0, 1, so, )
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); P2_SB
      P2_MM(1, 0, 0) P2_MM(1, 0, 1) P2_MM(1, 0, 2) P2_MM(1, 0, 3) P2_MM(1, 1, 0) P2_MM(1, 1, 1) P2_MM(1, 1, 2) P2_MM(1, 1, 3)
    }
#undef P2_SB
#undef P2_RA
#undef P2_RB
#undef P2_MM
#undef P2_GROUP_READING
    __builtin_amdgcn_s_barrier();                  // every wave has read its last fragments: all five slots are free
    // ---------------- epilogue: acc[i][j][4q + e] = C[mw + 32i + l31][nw + 32j + 8q + 4lh + e] ----------------
    const int mw = m0 + wm * 64, nw = GEGLU ? (wn == 0 ? tn * 128 : N + tn * 128) : n0 + wn * 128;
    const bool edge = m0 + 256 > M;
    const int vn = v + gridDim.x;
    const bool has_next = vn < nwg;
    if (has_next) {
      tile_ptrs(xcd_remap(vn, nwg));
      stage(0, 0); stage(32, 1); stage(64, 2);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      // bf16 payload of row block i: 32 rows x 256 B; 16-byte chunk c of row r at chunk c ^ (r & 15)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          u32x2v pk;
          pk[0] = pack2bf(acc[i][j][4 * q], acc[i][j][4 * q + 1]); pk[1] = pack2bf(acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
          const int ch = j * 4 + q;
          PS_DSW64(scratch + (unsigned)(l31 * 256 + ((ch ^ (l31 & 15)) << 4) + 8 * lh), pk);
        }
      u32x4v o[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int row = (lane >> 4) + 4 * u, ch = lane & 15;
        NT_DSREAD(o[u], scratch + (unsigned)(row * 256 + ((ch ^ (row & 15)) << 4)));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int m = mw + i * 32 + (lane >> 4) + 4 * u;
        u16* cp = C + (int64_t)m * ldc + nw + 8 * (lane & 15);
        if (m < M) PS_GSTORE(cp, o[u]);
      }
      if (GEGLU) {
        // g = a * gelu(gate): the partner wave (same rows, other column half) holds the other operand; both row blocks are
        // in the scratch areas now.  Wave wn takes rows 16*wn .. 16*wn + 15 of the pair's 32 rows.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned sa = lds_base + 3u * (unsigned)(P2_STAGE * 2) + (unsigned)(wave & ~1) * 8192u, sg = sa + 8192u;
        u32x4v av[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int row = 16 * wn + (lane >> 4) + 4 * u, ch = lane & 15;
          NT_DSREAD(av[u], sa + (unsigned)(row * 256 + ((ch ^ (row & 15)) << 4)));
          NT_DSREAD(gv[u], sg + (unsigned)(row * 256 + ((ch ^ (row & 15)) << 4)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int m = m0 + wm * 64 + i * 32 + 16 * wn + (lane >> 4) + 4 * u;
          u32x4v gg;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float a0 = __uint_as_float(av[u][e] << 16), a1 = __uint_as_float(av[u][e] & 0xffff0000u);
            const float g0 = __uint_as_float(gv[u][e] << 16), g1 = __uint_as_float(gv[u][e] & 0xffff0000u);
            float ge0, ge1, unused;
            gelu_pair(g0, ge0, unused); gelu_pair(g1, ge1, unused);
            gg[e] = pack2bf(a0 * ge0, a1 * ge1);
          }
          u16* gp = G + (int64_t)m * ldg + tn * 128 + 8 * (lane & 15);
          if (m < M) PS_GSTORE(gp, gg);
        }
        if (i == 0) asm volatile("s_barrier" ::: "memory");          // the partner has read this row block: scratch may be overwritten
      }
    }
    if (edge) ps_wait_vm<0>();
    first = false;
    v = vn;
  }
}

extern "C" int mca_gemm_nt(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                           int out_bf16, const float* bias, const float* residual, int64_t ldres,
                           int64_t res_period, int64_t M, int64_t N, int64_t K, mca_stream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return MCA_E_BADARG;
  if (K % 64 || lda % 8 || ldb % 8 || (uintptr_t)A % 16 || (uintptr_t)B % 16) return MCA_E_ALIGN;
  if (lda < K || ldb < K || ldc < N) return MCA_E_BADARG;
  if (M > (1LL << 30) || N > (1LL << 30)) return MCA_E_UNSUPPORTED;
  const int tiles_m = (int)((M + BM - 1) / BM), tiles_n = (int)((N + BN - 1) / BN);
  const int nwg = tiles_m * tiles_n;
  const int res = !residual ? 0 : (res_period > 0 ? 2 : 1);
  const bool big = M >= 2048 && g_knob[1] != 1;            // knob 1 = 1 forces the 128x128 kernel (A/B measurements)
  const int nwg2 = (int)((M + BM2 - 1) / BM2) * tiles_n;
#define NT_LAUNCH_G(OB, RS)                                                                                              \
  hipLaunchKernelGGL((gemm_nt_glds_kernel<OB, RS, 64, 0>), dim3(nwg), dim3(256), 0, as_stream(stream), A, lda, B, ldb, C,   \
                     ldc, bias, residual, ldres, res_period, (int)M, (int)N, (int)K, tiles_n, nwg)
#define NT_LAUNCH_256(OB, RS, PFV)                                                                                          \
  do {                                                                                                                   \
    static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);                                                                                            \
    if (!attr) {                                                                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_256_kernel<OB, RS, PFV, 0>),                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, NT256_LDS_BYTES) != hipSuccess)                \
        return MCA_E_LAUNCH;                                                                                             \
      attr = true;                                                                                                       \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm_nt_256_kernel<OB, RS, PFV, 0>), dim3(nwg2), dim3(512), NT256_LDS_BYTES, as_stream(stream), A, lda, \
                       B, ldb, C, ldc, bias, residual, ldres, res_period, (int)M, (int)N, (int)K, tiles_n, nwg2);        \
  } while (0)
#define NT_PICK_PF(OB, RS) do { if (big) NT_LAUNCH_256(OB, RS, 0); else NT_LAUNCH_G(OB, RS); } while (0)
  // persistent kernel for bf16 / plain fp32 outputs (knob 7, A/B measurements: 1 = one-tile-per-workgroup kernels only,
  // 3 = persistent kernel for fp32 + residual as well)
  const bool c16 = (uintptr_t)C % 16 == 0 && ldc % (out_bf16 ? 8 : 4) == 0;
  // (fp32 output + residual: HBM-bound, the lock-step kernel with its residual prefetch measures 10-16 % faster: MODE 2 of
  // the persistent kernel is only used with knob 7 = 3)
  const bool ps_res = res == 1 && !out_bf16 && ldres % 4 == 0 && (uintptr_t)residual % 16 == 0 && g_knob[7] == 3;
  // bf16 output, N % 256 == 0: 256x256 tiles (knob 10 = 1: keep the 256x128 persistent kernel, A/B)
  if (big && g_knob[7] == 0 && g_knob[10] != 1 && out_bf16 && res == 0 && !bias && N % 256 == 0 && K >= 192 && K % 32 == 0 && c16) {
    static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_persist256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              P2_LDS_BYTES) != hipSuccess)
        return MCA_E_LAUNCH;
      attr = true;
    }
    const int tn2 = (int)(N / 256), nw2 = (int)((M + 255) / 256) * tn2;
    const int grid = nw2 < num_cus() ? nw2 : num_cus();
    hipLaunchKernelGGL(gemm_nt_persist256_kernel<false>, dim3(grid), dim3(512), P2_LDS_BYTES, as_stream(stream), A, lda, B, ldb,
                       reinterpret_cast<u16*>(C), ldc, (u16*)nullptr, (int64_t)0, (int)M, (int)N, (int)K, tn2, nw2, g_knob[0]);
    return launch_status();
  }
  if (big && (g_knob[7] == 0 || g_knob[7] == 3) && N % BN == 0 && K >= 320 && c16 && (res == 0 || ps_res) && (!bias || (uintptr_t)bias % 16 == 0)) {
    const int grid = nwg2 < num_cus() ? nwg2 : num_cus();
#define NT_LAUNCH_PS(MODE, BI)                                                                                               \
  do {                                                                                                                   \
    static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);                                                                                            \
    if (!attr) {                                                                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_persist_kernel<MODE, BI>),                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, NTPS_LDS_BYTES) != hipSuccess)                 \
        return MCA_E_LAUNCH;                                                                                             \
      attr = true;                                                                                                       \
    }                                                                                                                    \
    hipLaunchKernelGGL((gemm_nt_persist_kernel<MODE, BI>), dim3(grid), dim3(512), NTPS_LDS_BYTES, as_stream(stream), A, lda, B, \
                       ldb, C, ldc, bias, residual, ldres, (int)M, (int)N, (int)K, tiles_n, nwg2, g_knob[0]);            \
  } while (0)
    if (out_bf16) { if (bias) NT_LAUNCH_PS(0, true); else NT_LAUNCH_PS(0, false); }
    else if (res == 0) { if (bias) NT_LAUNCH_PS(1, true); else NT_LAUNCH_PS(1, false); }
    else { if (bias) NT_LAUNCH_PS(2, true); else NT_LAUNCH_PS(2, false); }
    return launch_status();
  }
  const bool pf = big && !out_bf16 && res == 1 && N % BN == 0 && K >= 512 && ldres % 4 == 0 && ldc % 4 == 0 &&
                  (uintptr_t)residual % 16 == 0 && (uintptr_t)C % 16 == 0 && (!bias || (uintptr_t)bias % 4 == 0) && g_knob[4] != 1;
  if (pf) { NT_LAUNCH_256(false, 1, 1); return launch_status(); }
  if (out_bf16) { if (res == 0) NT_PICK_PF(true, 0); else if (res == 1) NT_PICK_PF(true, 1); else NT_PICK_PF(true, 2); }
  else { if (res == 0) NT_PICK_PF(false, 0); else if (res == 1) NT_PICK_PF(false, 1); else NT_PICK_PF(false, 2); }
  return launch_status();
}


// C[M,N] (fp32) = A·B^T + LayerNorm(x) with the LayerNorm recomputed in the epilogue from x and its saved statistics.
extern "C" int mca_gemm_nt_lnres(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                                 const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                 int64_t M, int64_t N, int64_t K, mca_stream_t stream) {
  if (!A || !B || !C || !x || !mean || !rstd || !gamma || M <= 0 || N <= 0 || K <= 0) return MCA_E_BADARG;
  if (K % 64 || lda % 8 || ldb % 8 || ldc % 4 || ldx % 4 || (uintptr_t)A % 16 || (uintptr_t)B % 16 || (uintptr_t)C % 16 ||
      (uintptr_t)x % 16 || (uintptr_t)gamma % 16)
    return MCA_E_ALIGN;
  if (lda < K || ldb < K || ldc < N || ldx < N) return MCA_E_BADARG;
  // the fused form exists for the large-M residual-prefetch kernel only; callers keep the two-kernel form otherwise
  if (M < 2048 || M > (1LL << 30) || N % BN || K < 512) return MCA_E_UNSUPPORTED;
  const int tiles_n = (int)(N / BN);
  const int nwg2 = (int)((M + BM2 - 1) / BM2) * tiles_n;
  static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_256_kernel<false, 1, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            NT256LN_LDS_BYTES) != hipSuccess)
      return MCA_E_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt_256_kernel<false, 1, 1, 2>), dim3(nwg2), dim3(512), NT256LN_LDS_BYTES, as_stream(stream), A, lda, B, ldb,
                     (void*)C, ldc, (const float*)nullptr, x, ldx, (int64_t)0, (int)M, (int)N, (int)K, tiles_n, nwg2, mean, rstd, gamma);
  return launch_status();
}

// dh = GEGLU'(h) applied to dg = A·B^T without materialising dg (fused epilogue).  A[M,K] (= d x_out, bf16), B[ip,K] (= W2^T
// copy), h / dh [M, 2*ip] bf16.
extern "C" int mca_gemm_nt_geglu_bwd(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, const uint16_t* h,
                                     uint16_t* dh, int64_t ldh, int64_t ip, int64_t M, int64_t K, mca_stream_t stream) {
  if (!A || !B || !h || !dh || M <= 0 || ip <= 0 || K <= 0) return MCA_E_BADARG;
  if (K % 64 || lda % 8 || ldb % 8 || ldh % 8 || ip % 8 || (uintptr_t)A % 16 || (uintptr_t)B % 16 || (uintptr_t)h % 16 || (uintptr_t)dh % 16)
    return MCA_E_ALIGN;
  if (lda < K || ldb < K || ldh < 2 * ip) return MCA_E_BADARG;
  if (M > (1LL << 30)) return MCA_E_UNSUPPORTED;
  const int64_t N = ip;
  const int tiles_n = (int)((N + BN - 1) / BN);
  const float* hres = reinterpret_cast<const float*>(h);
  void* C = dh;
  // below ~40k rows (the data-parallel configs' 8 samples per GPU: 20,304 rows) the 128 x 128 kernel, two or three workgroups
  // per CU whose epilogues (0.9 GB of h / dh traffic at b = 32) overlap each other's k-loops, beats the persistent 256 x 128
  // one (65 against 75 us at b = 8; 308 against 282 at b = 32: tools/bench_step_gemms.py); knob 1 = 1 forces it (A/B)
  if (M >= 40960 && g_knob[1] != 1) {
    const int nwg2 = (int)((M + BM2 - 1) / BM2) * tiles_n;
    static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_256_kernel<true, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              NT256_LDS_BYTES) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_persist_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              NTPS_LDS_BYTES) != hipSuccess)
        return MCA_E_LAUNCH;
      attr = true;
    }
    if (g_knob[7] != 1 && N % BN == 0 && K >= 320) {
      const int grid = nwg2 < num_cus() ? nwg2 : num_cus();
      hipLaunchKernelGGL((gemm_nt_persist_kernel<3, false>), dim3(grid), dim3(512), NTPS_LDS_BYTES, as_stream(stream), A, lda, B, ldb, C, ldh,
                         nullptr, hres, ldh, (int)M, (int)N, (int)K, tiles_n, nwg2, g_knob[0]);
    } else
      hipLaunchKernelGGL((gemm_nt_256_kernel<true, 0, 0, 1>), dim3(nwg2), dim3(512), NT256_LDS_BYTES, as_stream(stream), A, lda, B, ldb, C,
                         ldh, nullptr, hres, ldh, 0, (int)M, (int)N, (int)K, tiles_n, nwg2);
  } else {
    const int nwg = (int)((M + BM - 1) / BM) * tiles_n;
    hipLaunchKernelGGL((gemm_nt_glds_kernel<true, 0, 64, 1>), dim3(nwg), dim3(256), 0, as_stream(stream), A, lda, B, ldb, C, ldh, nullptr,
                       hres, ldh, 0, (int)M, (int)N, (int)K, tiles_n, nwg);
  }
  return launch_status();
}

// h = A·W1^T (bf16, [a | gate]) and g = a * gelu(gate) in one pass (persistent kernel MODE 4); small or oddly shaped
// problems take the plain GEMM followed by the element-wise kernel.
extern "C" int mca_gemm_nt_geglu_fwd(const uint16_t* A, int64_t lda, const uint16_t* W1, int64_t ldb, uint16_t* h, int64_t ldh,
                                     uint16_t* g, int64_t ldg, int64_t ip, int64_t M, int64_t K, mca_stream_t stream) {
  if (!A || !W1 || !h || !g || M <= 0 || ip <= 0 || K <= 0) return MCA_E_BADARG;
  if (K % 64 || lda % 8 || ldb % 8 || ldh % 8 || ldg % 8 || ip % 8 || (uintptr_t)A % 16 || (uintptr_t)W1 % 16 || (uintptr_t)h % 16 ||
      (uintptr_t)g % 16)
    return MCA_E_ALIGN;
  if (lda < K || ldb < K || ldh < 2 * ip || ldg < ip) return MCA_E_BADARG;
  if (M > (1LL << 30)) return MCA_E_UNSUPPORTED;
  if (M >= 2048 && ip % 128 == 0 && K >= 192 && g_knob[7] != 1 && g_knob[10] != 1) {
    static bool attr2_dev[64] = {false}; bool& attr2 = *mca_dev_flag(attr2_dev);
    if (!attr2) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_persist256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              P2_LDS_BYTES) != hipSuccess)
        return MCA_E_LAUNCH;
      attr2 = true;
    }
    const int tn2 = (int)(ip / 128), nw2 = (int)((M + 255) / 256) * tn2;
    const int grid = nw2 < num_cus() ? nw2 : num_cus();
    hipLaunchKernelGGL(gemm_nt_persist256_kernel<true>, dim3(grid), dim3(512), P2_LDS_BYTES, as_stream(stream), A, lda, W1, ldb, h, ldh, g,
                       ldg, (int)M, (int)ip, (int)K, tn2, nw2, g_knob[0]);
    return launch_status();
  }
  if (M >= 2048 && ip % 64 == 0 && K >= 320 && g_knob[7] != 1) {
    static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_persist_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              NTPS_LDS_BYTES) != hipSuccess)
        return MCA_E_LAUNCH;
      attr = true;
    }
    const int tiles_n = (int)(ip / 64);
    const int nwg2 = (int)((M + BM2 - 1) / BM2) * tiles_n;
    const int grid = nwg2 < num_cus() ? nwg2 : num_cus();
    hipLaunchKernelGGL((gemm_nt_persist_kernel<4, false>), dim3(grid), dim3(512), NTPS_LDS_BYTES, as_stream(stream), A, lda, W1, ldb, h, ldh,
                       nullptr, reinterpret_cast<const float*>(g), ldg, (int)M, (int)ip, (int)K, tiles_n, nwg2, g_knob[0]);
    return launch_status();
  }
  if (ldg != ip || ldh != 2 * ip) return MCA_E_UNSUPPORTED;          // the element-wise kernel takes packed rows
  const int rc = mca_gemm_nt(A, lda, W1, ldb, h, ldh, 1, nullptr, nullptr, 0, 0, M, 2 * ip, K, stream);
  if (rc != MCA_OK) return rc;
  return mca_geglu_fwd(h, g, M, (int)ip, stream);
}

// =====================================================================================================
// weight gradient:  C[N,K] += A[R,N]^T · B[R,K]
// LDS image of a [64 r][128 cols] tile (256-byte rows): 16-byte chunk c (0..15) of row r at chunk
// c ^ (4*(r & 3)): the four rows of one transposed 4x16 read then sit in four different 64-byte bank
// quarters (conflict-free ds_read_b64_tr_b16).
// =====================================================================================================
#define BR 64
__device__ __forceinline__ int tn_off(int r, int c) { return r * 128 + ((c ^ ((r & 3) << 2)) << 3); }

__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

__global__ __launch_bounds__(256) void gemm_tn_kernel(const u16* __restrict__ A, int64_t lda,
                                                       const u16* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                       int64_t ldc, int R, int N, int K, int tiles_k, int rows_per_split, int dbg) {
  __shared__ __attribute__((aligned(16))) u16 lds[2 * 2 * BR * 128];    // 64 KiB
  u16* As = lds;
  u16* Bs = lds + 2 * BR * 128;
  const int tn = blockIdx.x / tiles_k, tk = blockIdx.x % tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const int r_begin = blockIdx.y * rows_per_split;
  int r_end = r_begin + rows_per_split; if (r_end > R) r_end = R;
  if (r_begin >= r_end) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int lh = lane >> 5;

  // direct global->LDS staging: the tile image is 64 rows x 16 chunks (1024 chunks) per operand, 4 wave-instructions
  // per wave; LDS position p = (i*4 + wave)*64 + lane holds logical chunk (p & 15) ^ ((row & 3) << 2) of row p >> 4.
  // Rows past the end of this split and column chunks past N / K are fetched from a 16-byte zero block.
  int srow[4];
  const u16* pa[4];
  const u16* pb[4];
  bool ca_ok[4], cb_ok[4];
  const u16* zero = reinterpret_cast<const u16*>(g_zero16);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p = (i * 4 + wave) * 64 + lane, r = p >> 4, c = (p & 15) ^ ((r & 3) << 2);
    srow[i] = r;
    ca_ok[i] = n0 + c * 8 < N;
    cb_ok[i] = k0 + c * 8 < K;
    pa[i] = A + n0 + c * 8;
    pb[i] = B + k0 + c * 8;
  }
  auto stage = [&](int r0, int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + srow[i];
      const bool rok = r < r_end;
      const u16* sa = (rok && ca_ok[i]) ? pa[i] + (int64_t)r * lda : zero;
      const u16* sb = (rok && cb_ok[i]) ? pb[i] + (int64_t)r * ldb : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                       (__attribute__((address_space(3))) void*)(As + buf * BR * 128 + (i * 4 + wave) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                       (__attribute__((address_space(3))) void*)(Bs + buf * BR * 128 + (i * 4 + wave) * 512), 16, 0, 0);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // per-lane transposed-read coordinates: 16-lane group g = lane>>4 covers matrix rows 16*(g&1)..+15 of the
  // 32-row operand block and k-half lh; lane 4q+p supplies LDS row q, columns 4p..4p+3 of the 4x16 block.
  // The reads are issued as inline asm with hand-counted lgkmcnt: hipcc otherwise orders every
  // ds_read_b64_tr_b16 behind the LDS-DMA of the NEXT stage (s_waitcnt vmcnt(0) at the top of each step), which
  // serialises the DMA with the MFMAs.  Byte address of (ks, t, i): lane_base[i] + buf*16384 + ks*4096 + t*1024.
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  unsigned abase[2], bbase[2];
  {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u16*)lds;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int ca = wn * 64 + i * 32 + 16 * tg + 4 * tp, cb = wk * 64 + i * 32 + 16 * tg + 4 * tp;
      abase[i] = lds0 + 2u * (unsigned)((8 * lh + tq) * 128 + (((ca >> 3) ^ (tq << 2)) << 3) + (ca & 7));
      bbase[i] = lds0 + 2u * (unsigned)((8 * lh + tq) * 128 + (((cb >> 3) ^ (tq << 2)) << 3) + (cb & 7)) + 2u * 2 * BR * 128;
    }
  }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  const int nsteps = (r_end - r_begin + BR - 1) / BR;
  stage(r_begin, 0);
  __syncthreads();
  for (int st = 0; st < nsteps; st++) {
    const int cur = st & 1;
    if (st + 1 < nsteps) stage(r_begin + (st + 1) * BR, cur ^ 1);
    const unsigned a0 = abase[0] + cur * 16384u, a1 = abase[1] + cur * 16384u;
    const unsigned b0 = bbase[0] + cur * 16384u, b1 = bbase[1] + cur * 16384u;
    u32x2 fa[2][2][2], fb[2][2][2];          // [parity][i][t]
#define TN_ISSUE(KS, PAR)                                                        \
    TR_READ(fa[PAR][0][0], a0, (KS) * 4096); TR_READ(fa[PAR][0][1], a0, (KS) * 4096 + 1024); \
    TR_READ(fa[PAR][1][0], a1, (KS) * 4096); TR_READ(fa[PAR][1][1], a1, (KS) * 4096 + 1024); \
    TR_READ(fb[PAR][0][0], b0, (KS) * 4096); TR_READ(fb[PAR][0][1], b0, (KS) * 4096 + 1024); \
    TR_READ(fb[PAR][1][0], b1, (KS) * 4096); TR_READ(fb[PAR][1][1], b1, (KS) * 4096 + 1024);
#define TN_MFMA(PAR)                                                             \
    {                                                                            \
      bf16x8 af[2], bfr[2];                                                      \
      _Pragma("unroll") for (int i = 0; i < 2; i++) {                            \
        const uint4 ua = make_uint4(fa[PAR][i][0][0], fa[PAR][i][0][1], fa[PAR][i][1][0], fa[PAR][i][1][1]); \
        const uint4 ub = make_uint4(fb[PAR][i][0][0], fb[PAR][i][0][1], fb[PAR][i][1][0], fb[PAR][i][1][1]); \
        af[i] = *reinterpret_cast<const bf16x8*>(&ua);                           \
        bfr[i] = *reinterpret_cast<const bf16x8*>(&ub);                          \
      }                                                                          \
      _Pragma("unroll") for (int i = 0; i < 2; i++)                              \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                            \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0); \
    }
    TN_ISSUE(0, 0)
    TN_ISSUE(1, 1)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(0)
    __builtin_amdgcn_sched_barrier(0);
    TN_ISSUE(2, 0)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(1)
    __builtin_amdgcn_sched_barrier(0);
    TN_ISSUE(3, 1)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(1)
    __syncthreads();
  }
#undef TN_ISSUE
#undef TN_MFMA
#undef TR_READ
  // C[n][k]: row n in registers, column k on the lane -> 128-byte contiguous atomic segments per row
  const int l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int k = k0 + wk * 64 + j * 32 + l31;
    if (k >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < N && !(dbg & 1)) atomicAdd(C + (int64_t)n * ldc + k, acc[i][j][r]);
        if ((dbg & 1) && acc[i][j][r] == 123.456f) C[0] = 1.f;
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient, large variant: 256(n) x 128(k) output tile, 8 wavefronts (4x2, 64x64 each), three LDS stages of
// [64 r][256 n] + [64 r][128 k] (144 KiB) filled by global_load_lds two steps ahead, one raw s_barrier per step with a
// counted vmcnt (same pipeline as gemm_nt_256_kernel), transposed fragment reads in inline asm.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_off256(int r, int c) { return r * 256 + ((c ^ ((r & 3) << 2)) << 3); }

__global__ __launch_bounds__(512) void gemm_tn_256_kernel(const u16* __restrict__ A, int64_t lda,
                                                           const u16* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                           int64_t ldc, int R, int N, int K, int tiles_k, int rows_per_split, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 ldst[];
  constexpr int STAGE = BR * (256 + 128);        // elements per stage
  // XCD-aware order: the output tiles of ONE row split read the same rows of A and B, so they must share an L2.  The
  // linear workgroup id (XCD = id % 8) is remapped so that each XCD owns a contiguous range of (split, tile) pairs;
  // in launch order the tiles of a split were dealt round the eight XCDs and every XCD fetched every row (PMC: 905 MB
  // fetched per launch against ~330 MB of operands; 318 MB with the remap).
  const int lin0 = (int)(blockIdx.x + gridDim.x * blockIdx.y);
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y));          // knob 9 = 16: launch order (A/B)
  const int tile_id = lin % (int)gridDim.x, split_id = lin / (int)gridDim.x;
  const int tn = tile_id / tiles_k, tk = tile_id % tiles_k;
  const int n0 = tn * 256, k0 = tk * 128;
  const int r_begin = split_id * rows_per_split;
  int r_end = r_begin + rows_per_split; if (r_end > R) r_end = R;
  if (r_begin >= r_end) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int lh = lane >> 5;
  const u16* zero = reinterpret_cast<const u16*>(g_zero16);

  int arow[4], brow[2];
  const u16* pa[4];
  const u16* pb[2];
  bool ca_ok[4], cb_ok[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int p = (i * 8 + wave) * 64 + lane, r = p >> 5, c = (p & 31) ^ ((r & 3) << 2);
    arow[i] = r; ca_ok[i] = n0 + c * 8 < N; pa[i] = A + n0 + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = (i * 8 + wave) * 64 + lane, r = p >> 4, c = (p & 15) ^ ((r & 3) << 2);
    brow[i] = r; cb_ok[i] = k0 + c * 8 < K; pb[i] = B + k0 + c * 8;
  }
  auto stage = [&](int r0, int st) {
    u16* base = ldst + st * STAGE;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + arow[i];
      const u16* src = (r < r_end && ca_ok[i]) ? pa[i] + (int64_t)r * lda : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + (i * 8 + wave) * 512), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int r = r0 + brow[i];
      const u16* src = (r < r_end && cb_ok[i]) ? pb[i] + (int64_t)r * ldb : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + BR * 256 + (i * 8 + wave) * 512), 16, 0, 0);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  unsigned abase[2], bbase[2];
  {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u16*)ldst;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int ca = wn * 64 + i * 32 + 16 * tg + 4 * tp, cb = wk * 64 + i * 32 + 16 * tg + 4 * tp;
      abase[i] = lds0 + 2u * (unsigned)((8 * lh + tq) * 256 + (((ca >> 3) ^ (tq << 2)) << 3) + (ca & 7));
      bbase[i] = lds0 + 2u * (unsigned)(BR * 256) + 2u * (unsigned)((8 * lh + tq) * 128 + (((cb >> 3) ^ (tq << 2)) << 3) + (cb & 7));
    }
  }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  const int nsteps = (r_end - r_begin + BR - 1) / BR;
  stage(r_begin, 0);
  if (nsteps > 1) stage(r_begin + BR, 1);
  int st = 0;
  for (int sp = 0; sp < nsteps; sp++) {
    if (sp + 1 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (sp + 2 < nsteps) { int s2 = st + 2; if (s2 >= 3) s2 -= 3; stage(r_begin + (sp + 2) * BR, s2); }
    const unsigned so = (unsigned)st * (unsigned)(STAGE * 2);
    const unsigned a0 = abase[0] + so, a1 = abase[1] + so, b0 = bbase[0] + so, b1 = bbase[1] + so;
    u32x2 fa[2][2][2], fb[2][2][2];          // [parity][i][t]
#define TN_ISSUE(KS, PAR)                                                        \
    TR_READ(fa[PAR][0][0], a0, (KS) * 8192); TR_READ(fa[PAR][0][1], a0, (KS) * 8192 + 2048); \
    TR_READ(fa[PAR][1][0], a1, (KS) * 8192); TR_READ(fa[PAR][1][1], a1, (KS) * 8192 + 2048); \
    TR_READ(fb[PAR][0][0], b0, (KS) * 4096); TR_READ(fb[PAR][0][1], b0, (KS) * 4096 + 1024); \
    TR_READ(fb[PAR][1][0], b1, (KS) * 4096); TR_READ(fb[PAR][1][1], b1, (KS) * 4096 + 1024);
#define TN_MFMA(PAR)                                                             \
    {                                                                            \
      bf16x8 af[2], bfr[2];                                                      \
      _Pragma("unroll") for (int i = 0; i < 2; i++) {                            \
        const uint4 ua = make_uint4(fa[PAR][i][0][0], fa[PAR][i][0][1], fa[PAR][i][1][0], fa[PAR][i][1][1]); \
        const uint4 ub = make_uint4(fb[PAR][i][0][0], fb[PAR][i][0][1], fb[PAR][i][1][0], fb[PAR][i][1][1]); \
        af[i] = *reinterpret_cast<const bf16x8*>(&ua);                           \
        bfr[i] = *reinterpret_cast<const bf16x8*>(&ub);                          \
      }                                                                          \
      _Pragma("unroll") for (int i = 0; i < 2; i++)                              \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                            \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0); \
    }
    TN_ISSUE(0, 0)
    TN_ISSUE(1, 1)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(0)
    __builtin_amdgcn_sched_barrier(0);
    TN_ISSUE(2, 0)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(1)
    __builtin_amdgcn_sched_barrier(0);
    TN_ISSUE(3, 1)
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(0)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
    TN_MFMA(1)
    st = st == 2 ? 0 : st + 1;
  }
#undef TN_ISSUE
#undef TN_MFMA
#undef TR_READ
  const int l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int k = k0 + wk * 64 + j * 32 + l31;
    if (k >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < N) atomicAdd(C + (int64_t)n * ldc + k, acc[i][j][r]);
      }
  }
}
#define TN256_LDS_BYTES (3 * BR * (256 + 128) * 2)

// ---------------------------------------------------------------------------------------------------------
// Weight gradient, 256(n) x 256(k) output tile: the k-loop of these GEMMs is bound by the L2 -> LDS path per CU
// (DESIGN.md section 5), so the lever is bytes per flop: 32 KiB of operands per 32-row step feed 16 MFMAs per
// wavefront, against 24 KiB for the same 16 MFMAs... per 2x the flops: 256x256 moves 2/3 of the bytes per flop of
// 256x128.  8 wavefronts as 4 (n) x 2 (k), each 64 x 128 = 2 x 4 accumulators (128 VGPRs); FIVE LDS stages of
// 32 rows (160 KiB), DMA four steps ahead; fragments by ds_read_b64_tr_b16 (12 per k16-step for 8 MFMAs).
// ---------------------------------------------------------------------------------------------------------
#define BR2 32
#define TN2_STAGE (BR2 * 512)          // elements per stage: A image [32][256] then B image [32][256]
#define TN2_NSTAGE 5
#define TN2_LDS_BYTES (TN2_NSTAGE * TN2_STAGE * 2)
__device__ __forceinline__ void tn_256x256_tile(const u16* __restrict__ A, int64_t lda, const u16* __restrict__ B, int64_t ldb,
                                                float* __restrict__ C, int64_t ldc, int N, int K, int tn, int tk, int r_begin,
                                                int r_end, u16* ldst) {
  const int n0 = tn * 256, k0 = tk * 256;
  if (r_begin >= r_end) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int lh = lane >> 5;
  const u16* zero = reinterpret_cast<const u16*>(g_zero16);

  // DMA: an operand image is 1024 chunks of 16 B = 16 wave-instructions, two per wave; position p -> row p >> 5, chunk
  // position p & 31 holds logical chunk (p & 31) ^ ((row & 3) << 2).  Every lane keeps a running source pointer per load (rows
  // advance by 32 per stage), so a stage whose 32 rows all exist
  // costs four loads and four 64-bit adds; only a split's last, partial stage tests rows.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int srow[2];
  const u16* cur_a[2];
  const u16* cur_b[2];
  const int64_t inc_a = (int64_t)BR2 * lda, inc_b = (int64_t)BR2 * ldb;
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = (i * 8 + wave) * 64 + lane, r = p >> 5, c = (p & 31) ^ ((r & 3) << 2);
    srow[i] = r;
    // a column block past the matrix re-reads the last one inside it: those products land in rows / columns of C that are not stored
    int ca = n0 + c * 8, cb = k0 + c * 8;
    if (ca >= N) ca = (N - 1) & ~7;
    if (cb >= K) cb = (K - 1) & ~7;
    cur_a[i] = A + ca + (int64_t)(r_begin + r) * lda;
    cur_b[i] = B + cb + (int64_t)(r_begin + r) * ldb;
  }
  int r_next = r_begin;          // first row of the next stage to be issued (stages are issued in row order)
  auto stage = [&](int st) {
    u16* base = ldst + st * TN2_STAGE;
    const bool whole = r_next + BR2 <= r_end;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bool in = whole || r_next + srow[i] < r_end;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in ? cur_a[i] : zero),
                                       (__attribute__((address_space(3))) void*)(base + (i * 8 + wave_u) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in ? cur_b[i] : zero),
                                       (__attribute__((address_space(3))) void*)(base + BR2 * 256 + (i * 8 + wave_u) * 512), 16, 0, 0);
      cur_a[i] += inc_a; cur_b[i] += inc_b;
    }
    r_next += BR2;
  };
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // transposed-read coordinates as in gemm_tn_256_kernel: lane supplies LDS row 8*lh + tq (+4 for read t = 1, +16 per
  // k16-step), columns 16*tg + 4*tp .. +3 of its 32-column operand block
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  unsigned abase[2], bbase[4];
  {
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u16*)ldst;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int ca = wn * 64 + i * 32 + 16 * tg + 4 * tp;
      abase[i] = lds0 + 2u * (unsigned)((8 * lh + tq) * 256 + (((ca >> 3) ^ (tq << 2)) << 3) + (ca & 7));
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int cb = wk * 128 + j * 32 + 16 * tg + 4 * tp;
      bbase[j] = lds0 + 2u * (unsigned)(BR2 * 256) + 2u * (unsigned)((8 * lh + tq) * 256 + (((cb >> 3) ^ (tq << 2)) << 3) + (cb & 7));
    }
  }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
  // FIVE stages of 32 rows (all 160 KiB of LDS), DMA four steps ahead.  The step is software-pipelined: the fragments of one
  // k16-step are read in the gaps between the MFMAs of the k16-step before it (two reads per gap, none in a burst), and the step's
  // one barrier sits between its two MFMA groups, followed by one MFMA before the next stage's DMA is issued.
  const int nsteps = (r_end - r_begin + BR2 - 1) / BR2;
  stage(0);
  if (nsteps > 1) stage(1);
  if (nsteps > 2) stage(2);
  if (nsteps > 3) stage(3);
  u32x2 fa[2][2][2], fb[2][4][2];          // [k16-step][block][t]
#define TN2_SB __builtin_amdgcn_sched_barrier(0);
#define TN2_RA(KS, I, SO) TR_READ(fa[KS][I][0], abase[I] + (SO), (KS) * 8192); TR_READ(fa[KS][I][1], abase[I] + (SO), (KS) * 8192 + 2048); TN2_SB
#define TN2_RB(KS, J, SO) TR_READ(fb[KS][J][0], bbase[J] + (SO), (KS) * 8192); TR_READ(fb[KS][J][1], bbase[J] + (SO), (KS) * 8192 + 2048); TN2_SB
#define TN2_AF(KS, I) __builtin_bit_cast(bf16x8, make_uint4(fa[KS][I][0][0], fa[KS][I][0][1], fa[KS][I][1][0], fa[KS][I][1][1]))
#define TN2_BF(KS, J) __builtin_bit_cast(bf16x8, make_uint4(fb[KS][J][0][0], fb[KS][J][0][1], fb[KS][J][1][0], fb[KS][J][1][1]))
#define TN2_MM(KS, I, J) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TN2_AF(KS, I), TN2_BF(KS, J), acc[I][J], 0, 0, 0); TN2_SB
  // the MFMAs of k16-step KS with the reads of k16-step KN (at stage offset SO) in their gaps; MID runs after the first MFMA
#define TN2_GROUP_READING(KS, KN, SO, MID)                                                              \
    TN2_MM(KS, 0, 0) MID TN2_RA(KN, 0, SO) TN2_MM(KS, 0, 1) TN2_RB(KN, 0, SO) TN2_MM(KS, 0, 2) TN2_RB(KN, 1, SO) \
    TN2_MM(KS, 0, 3) TN2_RB(KN, 2, SO) TN2_MM(KS, 1, 0) TN2_RB(KN, 3, SO) TN2_MM(KS, 1, 1) TN2_RA(KN, 1, SO)    \
    TN2_MM(KS, 1, 2) TN2_MM(KS, 1, 3)
#define TN2_GROUP(KS)                                                                                   \
    TN2_MM(KS, 0, 0) TN2_MM(KS, 0, 1) TN2_MM(KS, 0, 2) TN2_MM(KS, 0, 3) TN2_MM(KS, 1, 0) TN2_MM(KS, 1, 1) TN2_MM(KS, 1, 2) TN2_MM(KS, 1, 3)
  if (nsteps > 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (nsteps > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (nsteps > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  TN2_SB
  TN2_RA(0, 0, 0u) TN2_RB(0, 0, 0u) TN2_RB(0, 1, 0u) TN2_RB(0, 2, 0u) TN2_RB(0, 3, 0u) TN2_RA(0, 1, 0u)
  int st = 0, prev = TN2_NSTAGE - 1;
  for (int sp = 0; sp + 1 < nsteps; sp++) {
    const unsigned so = (unsigned)st * (unsigned)(TN2_STAGE * 2);
    const int nxt = st == TN2_NSTAGE - 1 ? 0 : st + 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TN2_SB
    TN2_GROUP_READING(0, 1, so, )
    // stage sp + 1 has landed once only the (4 loads each of the) two younger stages are outstanding; past the barrier every
    // wavefront has finished with stage sp - 1, whose buffer takes stage sp + 4
    if (sp + 3 < nsteps) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (sp + 2 < nsteps) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    TN2_SB
    const unsigned son = (unsigned)nxt * (unsigned)(TN2_STAGE * 2);
    const bool fetch = sp + 4 < nsteps;
    TN2_GROUP_READING(1, 0, son, if (fetch) stage(prev); TN2_SB)
    prev = st; st = nxt;
  }
  {          // the last stage: nothing left to wait for or to fetch
    const unsigned so = (unsigned)st * (unsigned)(TN2_STAGE * 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TN2_SB
    TN2_GROUP_READING(0, 1, so, )
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TN2_SB
    TN2_GROUP(1)
  }
#undef TN2_SB
#undef TN2_RA
#undef TN2_RB
#undef TN2_AF
#undef TN2_BF
#undef TN2_MM
#undef TN2_GROUP_READING
#undef TN2_GROUP
#undef TR_READ
  // C[n][k]: row n in registers, column k on the lane -> 128-byte contiguous atomic segments per row
  const int l31 = lane & 31;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = k0 + wk * 128 + j * 32 + l31;
    if (k >= K) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < N) atomicAdd(C + (int64_t)n * ldc + k, acc[i][j][r]);
      }
  }
}

__global__ __launch_bounds__(512) void gemm_tn_256x256_kernel(const u16* __restrict__ A, int64_t lda,
                                                               const u16* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                               int64_t ldc, int R, int N, int K, int tiles_k, int rows_per_split, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 ldst[];
  const int lin0 = (int)(blockIdx.x + gridDim.x * blockIdx.y);
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y));          // tiles of one split on one XCD
  const int tile_id = lin % (int)gridDim.x, split_id = lin / (int)gridDim.x;
  const int r_begin = split_id * rows_per_split;
  int r_end = r_begin + rows_per_split; if (r_end > R) r_end = R;
  tn_256x256_tile(A, lda, B, ldb, C, ldc, N, K, tile_id / tiles_k, tile_id % tiles_k, r_begin, r_end, ldst);
}

// Several weight gradients over the SAME token rows in one launch (the four of a transformer layer): the launch then has
// 48 tiles instead of 12, so a full round of workgroups needs 5 row splits instead of 21 and the fp32 atomic traffic (every
// split adds the whole gradient once; a quarter to a third of the single launches' time) drops four-fold.
#define TN_SPAN_RELIEF 1536       // rows a two-tile workgroup is relieved of (mca_gemm_tn_acc_group; measured: tools/bench_tn_group.py)
struct tn_group {
  const u16* A[MCA_TN_MAX_GROUP];
  const u16* B[MCA_TN_MAX_GROUP];
  float* C[MCA_TN_MAX_GROUP];
  int64_t lda[MCA_TN_MAX_GROUP], ldb[MCA_TN_MAX_GROUP], ldc[MCA_TN_MAX_GROUP];
  int N[MCA_TN_MAX_GROUP], K[MCA_TN_MAX_GROUP], tiles_k[MCA_TN_MAX_GROUP];
  int first_tile[MCA_TN_MAX_GROUP + 1];
  int n, R, tiles;
  // Balanced row partition: every workgroup reduces `unit` rows of one tile's worth of work.  The first n_full * tiles
  // workgroups take whole (tile, split) cells of `unit` rows as before (the tiles of one split next to each other on an XCD, so
  // that an operand row block is fetched into one L2 once); the rows left over, [n_full * unit, R) of every tile, form a second,
  // tile-major line of tiles * (R - n_full * unit) row-units that the remaining workgroups cut into equal spans of `span` rows (a little less than `unit`): such a
  // workgroup finishes one tile's rest and starts the next one's (two atomic epilogues).  Any number of tiles then fills the
  // chip's one round of workgroups: 52 tiles are 4 full splits on 208 CUs + 48 spans, not 4 splits with 48 CUs idle (worth 7 % at
  // b = 32, not 19 %: the launch is bound by the shared L2 -> LDS and atomic paths, tools/bench_tn_group.py).
  int unit, n_full, span;          // rows of a whole cell, cells per tile, rows of a span of the line
  // own > 0 (when the line's workgroups are at least half as many as the tiles): the first `own` of them each take the whole
  // rest of "their" tile first - rows [n_full * unit, R) of tiles 0 .. own - 1, the SAME rows at the same time, so these segments
  // share operand rows through L2 like the cells do - and only tiles own .. tiles - 1 form the line (spans of `span` rows)
  int own;
};
__global__ __launch_bounds__(512) void gemm_tn_256x256_group_kernel(tn_group g, int dbg) {
  extern __shared__ __attribute__((aligned(16))) u16 ldst[];
  const int lin0 = (int)blockIdx.x;
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)gridDim.x);          // tiles of one split on one XCD
#ifdef MCA_TRACE_BUILD          // clock probe (knob 9 bit 8): shader cycles and 100 MHz ticks of every workgroup's lifetime (tools/ablate_tn_group.py)
  const bool probe = (dbg & 8) && threadIdx.x == 0 && lin0 < 512;
  const uint64_t c0 = probe ? __builtin_amdgcn_s_memtime() : 0, t0 = probe ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
  const int n_cells = g.n_full * g.tiles;
  const int row0 = g.n_full * g.unit, rest = g.R - row0;          // rows [row0, R) of every tile
  // this workgroup's span [s, e) of the tile-major line over tiles line0 .. tiles - 1 (a whole cell is the degenerate case: one
  // segment, no line arithmetic)
  const int line0 = g.own;
  int64_t s = 0, e = 0;
  bool own_pending = false;
  if (lin >= n_cells) {
    const int j = lin - n_cells;
    own_pending = j < g.own;
    s = (int64_t)j * g.span;
    e = s + g.span;
    const int64_t line = (int64_t)(g.tiles - line0) * rest;
    if (e > line) e = line;
  }
  bool first = true;
  for (;;) {
    int tile_all, r_begin, r_end;
    if (lin < n_cells) {
      tile_all = lin % g.tiles;
      r_begin = (lin / g.tiles) * g.unit; r_end = r_begin + g.unit;
    } else if (own_pending) {
      own_pending = false;
      tile_all = lin - n_cells;
      r_begin = row0; r_end = g.R;
    } else {
      if (s >= e) break;
      const int t = (int)(s / rest);
      tile_all = line0 + t;
      const int off = (int)(s - (int64_t)t * rest);
      int len = rest - off; if ((int64_t)len > e - s) len = (int)(e - s);
      r_begin = row0 + off; r_end = r_begin + len;
      s += len;
    }
    if (r_end > g.R) r_end = g.R;
    int p = 0;
#pragma unroll
    for (int i = 1; i < MCA_TN_MAX_GROUP; i++) if (i < g.n && tile_all >= g.first_tile[i]) p = i;
    const int tile_id = tile_all - g.first_tile[p];
    if (!first) __syncthreads();          // the previous segment's last fragment reads, before this one's DMA lands in the ring
    first = false;
    tn_256x256_tile(g.A[p], g.lda[p], g.B[p], g.ldb[p], g.C[p], g.ldc[p], g.N[p], g.K[p], tile_id / g.tiles_k[p], tile_id % g.tiles_k[p],
                    r_begin, r_end, ldst);
    if (lin < n_cells) break;
  }
#ifdef MCA_TRACE_BUILD
  if (probe) { mca_trace_gemm[2 * lin0] = __builtin_amdgcn_s_memtime() - c0; mca_trace_gemm[2 * lin0 + 1] = __builtin_amdgcn_s_memrealtime() - t0; }
#endif
}

extern "C" int mca_gemm_tn_acc(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                               int64_t R, int64_t N, int64_t K, mca_stream_t stream) {
  if (!A || !B || !C || R <= 0 || N <= 0 || K <= 0) return MCA_E_BADARG;
  if (lda % 8 || ldb % 8 || (uintptr_t)A % 16 || (uintptr_t)B % 16) return MCA_E_ALIGN;
  if (lda < (N + 7) / 8 * 8 || ldb < (K + 7) / 8 * 8 || ldc < K) return MCA_E_BADARG;
  if (R > (1LL << 30)) return MCA_E_UNSUPPORTED;
  const bool big = N >= 512 && R >= 4096 && g_knob[5] != 1;          // knob 5 = 1 forces the 128x128 kernel, 2 the 256x128 one
  // 256x256 tiles when the output has at least 8 of them (a 512x512 gradient has 4: the 256x128 kernel with half the
  // splits, i.e. half the atomic bytes, measured 70 vs 93 us)
  const bool huge = big && K >= 512 && g_knob[5] != 2 && ((N + 255) / 256) * ((K + 255) / 256) >= 8;
  const int tiles_k = (int)(huge ? (K + 255) / 256 : (K + 127) / 128);
  const int tiles_n = big ? (int)((N + 255) / 256) : (int)((N + 127) / 128);
  const int tiles = tiles_n * tiles_k;
  // split the reduction: one (big: 1 WG/CU) or two (2 WGs/CU) full rounds of workgroups; every split adds N*K*4 bytes
  // of fp32 atomics; at least 4 steps of 64 rows each
  int64_t splits = big ? (tiles <= 16 ? 256 / tiles : 512 / tiles) : (tiles <= 32 ? 512 / tiles : 1024 / tiles);
  if (huge) splits = 256 / tiles > 0 ? 256 / tiles : 1;
  if (g_knob[3] > 0) splits = g_knob[3];
  const int64_t max_splits = (R + 4 * BR - 1) / (4 * BR);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  int64_t rps = (R + splits - 1) / splits;
  rps = (rps + BR - 1) / BR * BR;
  splits = (R + rps - 1) / rps;
  if (huge) {
    static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_256x256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              TN2_LDS_BYTES) != hipSuccess)
        return MCA_E_LAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL(gemm_tn_256x256_kernel, dim3(tiles, (unsigned)splits), dim3(512), TN2_LDS_BYTES, as_stream(stream), A, lda, B,
                       ldb, C, ldc, (int)R, (int)N, (int)K, tiles_k, (int)rps, g_knob[9]);
    return launch_status();
  }
  if (big) {
    static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_256_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              TN256_LDS_BYTES) != hipSuccess)
        return MCA_E_LAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL(gemm_tn_256_kernel, dim3(tiles, (unsigned)splits), dim3(512), TN256_LDS_BYTES, as_stream(stream), A, lda, B,
                       ldb, C, ldc, (int)R, (int)N, (int)K, tiles_k, (int)rps, g_knob[9]);
    return launch_status();
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, (unsigned)splits), dim3(256), 0, as_stream(stream), A, lda, B, ldb, C,
                     ldc, (int)R, (int)N, (int)K, tiles_k, (int)rps, g_knob[2]);
  return launch_status();
}

extern "C" int mca_gemm_tn_acc_group(const mca_tn_desc* d, int n, int64_t R, mca_stream_t stream) {
  if (!d || n <= 0 || n > MCA_TN_MAX_GROUP || R <= 0) return MCA_E_BADARG;
  if (R > (1LL << 30)) return MCA_E_UNSUPPORTED;
  // The grouped kernel is the 256x256-tile one: members it does not suit (or a group too small to fill the chip) go
  // through the single-problem entry point, one launch each.
  tn_group g;
  int tiles = 0;
  bool ok = g_knob[11] != 1 && R >= 4096;          // knob 11 = 1: always one launch per member (A/B)
  for (int i = 0; i < n && ok; i++) {
    if (!d[i].A || !d[i].B || !d[i].C || d[i].N <= 0 || d[i].K <= 0) return MCA_E_BADARG;
    if (d[i].lda % 8 || d[i].ldb % 8 || (uintptr_t)d[i].A % 16 || (uintptr_t)d[i].B % 16) return MCA_E_ALIGN;
    if (d[i].lda < (d[i].N + 7) / 8 * 8 || d[i].ldb < (d[i].K + 7) / 8 * 8 || d[i].ldc < d[i].K) return MCA_E_BADARG;
    if (d[i].N < 256 || d[i].K < 256 || d[i].N > (1 << 24) || d[i].K > (1 << 24)) { ok = false; break; }
    g.A[i] = d[i].A; g.B[i] = d[i].B; g.C[i] = d[i].C;
    g.lda[i] = d[i].lda; g.ldb[i] = d[i].ldb; g.ldc[i] = d[i].ldc;
    g.N[i] = (int)d[i].N; g.K[i] = (int)d[i].K;
    g.tiles_k[i] = (int)((d[i].K + 255) / 256);
    g.first_tile[i] = tiles;
    tiles += (int)((d[i].N + 255) / 256) * g.tiles_k[i];
  }
  if (!ok || n == 1 || tiles < 16 || tiles > 65535) {
    for (int i = 0; i < n; i++) {
      const int rc = mca_gemm_tn_acc(d[i].A, d[i].lda, d[i].B, d[i].ldb, d[i].C, d[i].ldc, R, d[i].N, d[i].K, stream);
      if (rc != MCA_OK) return rc;
    }
    return MCA_OK;
  }
  for (int i = n; i <= MCA_TN_MAX_GROUP; i++) g.first_tile[i] = tiles;
  g.n = n; g.R = (int)R; g.tiles = tiles;
  // One full round of workgroups (1 per CU).  n_full whole splits of `unit` rows per tile + `spans` workgroups on the tile-major
  // line of the rows left over (struct tn_group).  A span workgroup pays two pipeline fills and two atomic epilogues, so it gets
  // `relief` rows less than a cell: unit = (tiles * R + spans * relief) / CUs.  At least 4 steps of 32 rows per workgroup.
  // knob 3 = s: s uniform splits and no line (the round-3 partition, A/B); knob 6 = r + 1: relief of 32 r rows, -(r + 1): the same without owner segments
  const int cus = num_cus();
  const int k6 = g_knob[6];
  // (a FIXED relief: scaling it down with R - at most an eighth of a cell - was measured and is worse at small R: b = 2, 52 / 60 tiles
  //  73.6 / 78.0 us with 1,536 rows against 76.9 / 95.0 with 512; b = 16 within 2 % either way.  tools/bench_tn_group.py 2 8 16)
  const int64_t relief = k6 != 0 ? 32 * (int64_t)((k6 < 0 ? -k6 : k6) - 1) : TN_SPAN_RELIEF;
  int64_t n_full = cus / tiles, unit, rest, span = 0, spans = 0, own = 0;
  if (g_knob[3] > 0) {
    unit = ((R + g_knob[3] - 1) / g_knob[3] + BR2 - 1) / BR2 * BR2;
    n_full = (R + unit - 1) / unit; rest = 0;                 // uniform splits: the last one is short
  } else {
    const int64_t sp0 = cus - n_full * tiles, left = tiles - sp0;          // line workgroups; tiles without an owner among them
    const bool owners = k6 >= 0 && n_full > 0 && sp0 > 0 && left > 0 && left <= sp0;
    if (owners) {
      // a line workgroup reduces rest + left * rest / sp0 rows, a cell `unit` = that + relief:  rest * f + relief = unit,
      // rest = R - n_full * unit,  f = 1 + left / sp0
      const double f = 1.0 + (double)left / (double)sp0;
      unit = (int64_t)(((double)R * f + (double)relief) / ((double)n_full * f + 1.0));
      if (unit < 4 * BR2) unit = 4 * BR2;
      unit = (unit + BR2 - 1) / BR2 * BR2;
      rest = R - n_full * unit;
    }
    if (owners && rest >= BR2) {
      own = sp0;
      span = (left * rest + sp0 - 1) / sp0;
      if (span < 4 * BR2) span = 4 * BR2;
      span = (span + BR2 - 1) / BR2 * BR2;
      spans = sp0;
    } else {
      for (;;) {
        const int64_t sp = cus - n_full * tiles;                // workgroups left for the line
        unit = ((int64_t)tiles * R + sp * relief + cus - 1) / cus;
        if (unit < 4 * BR2) unit = 4 * BR2;
        unit = (unit + BR2 - 1) / BR2 * BR2;
        rest = R - n_full * unit;
        if (rest >= 0 || n_full == 0) break;
        n_full--;                                               // (tiny R: fewer whole splits)
      }
      if (n_full == 0) { rest = R; }
      if (rest > 0) {
        const int64_t sp = cus - n_full * tiles > 0 ? cus - n_full * tiles : cus;
        span = ((int64_t)tiles * rest + sp - 1) / sp;
        if (span < 4 * BR2) span = 4 * BR2;
        span = (span + BR2 - 1) / BR2 * BR2;
        spans = ((int64_t)tiles * rest + span - 1) / span;
      }
    }
  }
  const int64_t grid = n_full * tiles + spans;
  if (grid <= 0 || grid > (1 << 30)) return MCA_E_UNSUPPORTED;
  g.span = (int)span; g.own = (int)own;
  g.unit = (int)unit; g.n_full = (int)n_full;
  static bool attr_dev[64] = {false}; bool& attr = *mca_dev_flag(attr_dev);
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_256x256_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            TN2_LDS_BYTES) != hipSuccess)
      return MCA_E_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(gemm_tn_256x256_group_kernel, dim3((unsigned)grid), dim3(512), TN2_LDS_BYTES, as_stream(stream), g, g_knob[9]);
  return launch_status();
}

// bf16 MFMA GEMMs for every Linear of the MCA step (forward, data-gradient and weight-gradient).
//
//   mca_gemm_nt     C[M,N]  = A[M,K] · B[N,K]^T      both operands K-contiguous (Linear forward with
//                                                     x[M,K], W[N,K]; data-gradient with the transposed
//                                                     bf16 weight copy)
//   mca_gemm_tn_acc C[N,K] += A[R,N]^T · B[R,K]      both operands reduction-major (weight gradient:
//                                                     dW = dY^T · X), operands fetched from LDS with the
//                                                     gfx950 transposed read ds_read_b64_tr_b16
//
// Tile 128x128x64, 256 threads = 4 wavefronts (2x2), each wavefront a 64x64 sub-tile as 2x2
// v_mfma_f32_32x32x16_bf16 accumulators.  NT: operand tiles go global -> LDS directly
// (global_load_lds_dwordx4, double-buffered) with the XOR swizzle on the source address and on the fragment
// reads (16-byte reads bank-conflict free); the C tile leaves through LDS as whole 16-byte row pieces with
// bias / residual fused.  TN: register-staged tiles, transposed LDS reads, fp32 atomics.  Workgroup ids are
// remapped so that the blocks sharing an XCD (ids congruent mod 8) walk neighbouring tiles and reuse the
// A panel from that XCD's L2.
#include "common.h"

#define BM 128
#define BN 128
#define BK 64

// ---- XCD-aware bijective remap of a linear block id (guide §5 "XCD swizzle must be bijective") ----
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / 8;
}

// LDS image of a [128 rows][64 bf16] tile (128-byte rows): 16-byte chunk c of row r lives at chunk
// c ^ ((r >> 1) & 7): a 16-lane group of ds_read_b128 (16 distinct rows, same logical chunk) then covers
// all 64 banks exactly once.
__device__ __forceinline__ int nt_off(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 7)) << 3); }

// Epilogue through LDS (BK = 64 kernels: the 64 KiB of operand buffers hold exactly one 128x128 fp32 tile).  The
// accumulators are written lane = column / register = row, then every thread reads whole 16-byte pieces of rows,
// adds bias / residual with 16-byte loads and stores 16 bytes: 4x (fp32) or 8x (bf16) fewer store instructions,
// every global access a full 128-byte-line segment of a row.
template <bool OUT_BF16, int RES>
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
  constexpr int EPT = OUT_BF16 ? 8 : 4;          // elements per thread per piece
  constexpr int PPR = 128 / EPT;                 // pieces per row
#pragma unroll 4
  for (int it = 0; it < (128 * PPR) / 256; it++) {
    const int id = tid + 256 * it;
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

// ---------------------------------------------------------------------------------------------------------
// NT kernel with direct global->LDS staging (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass.
// The LDS destination of one wave-instruction is linear (wave base + lane*16 B), so the XOR swizzle is applied
// to the per-lane SOURCE address and again on the fragment reads (cdna guide, rule 21).
// ---------------------------------------------------------------------------------------------------------
template <int BKT> __device__ __forceinline__ int gl_sw(int r) { return BKT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
template <int BKT> __device__ __forceinline__ int gl_off(int r, int c) { return r * BKT + ((c ^ gl_sw<BKT>(r)) << 3); }

template <bool OUT_BF16, int RES, int BKT>
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
  stage(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nkt; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage((kt + 1) * BKT, cur ^ 1);
    const u16* as = As + cur * TILE;
    const u16* bs = Bs + cur * TILE;
#pragma unroll
    for (int ks = 0; ks < BKT / 16; ks++) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        af[i] = *reinterpret_cast<const bf16x8*>(as + gl_off<BKT>(wm * 64 + i * 32 + l31, 2 * ks + lh));
        bfr[i] = *reinterpret_cast<const bf16x8*>(bs + gl_off<BKT>(wn * 64 + i * 32 + l31, 2 * ks + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();          // waits for this wave's LDS-DMA (vmcnt(0)) and for every wave's reads of `cur`
  }
  static_assert(BKT == 64, "the LDS epilogue needs the 64 KiB of operand buffers");
  nt_epilogue_lds<OUT_BF16, RES>(acc, reinterpret_cast<float*>(lds), Cv, ldc, bias, residual, ldres, res_period, M, N, m0, n0,
                                 wm, wn, l31, lh, tid);
}

static int g_knob[8];
extern "C" int mca_debug_set(int key, int value) { if (key >= 0 && key < 8) g_knob[key] = value; return 0; }

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
#define NT_LAUNCH_G(OB, RS)                                                                                              \
  hipLaunchKernelGGL((gemm_nt_glds_kernel<OB, RS, 64>), dim3(nwg), dim3(256), 0, as_stream(stream), A, lda, B, ldb, C,   \
                     ldc, bias, residual, ldres, res_period, (int)M, (int)N, (int)K, tiles_n, nwg)
#define NT_PICK_PF(OB, RS) NT_LAUNCH_G(OB, RS)
  if (out_bf16) { if (res == 0) NT_PICK_PF(true, 0); else if (res == 1) NT_PICK_PF(true, 1); else NT_PICK_PF(true, 2); }
  else { if (res == 0) NT_PICK_PF(false, 0); else if (res == 1) NT_PICK_PF(false, 1); else NT_PICK_PF(false, 2); }
  return launch_status();
}

// =====================================================================================================
// weight gradient:  C[N,K] += A[R,N]^T · B[R,K]
// LDS image of a [64 r][128 cols] tile (256-byte rows): 16-byte chunk c (0..15) of row r at chunk
// c ^ (4*(r & 3)): the four rows of one transposed 4x16 read then sit in four different 64-byte bank
// quarters (conflict-free ds_read_b64_tr_b16).
// =====================================================================================================
#define BR 64
__device__ __forceinline__ int tn_off(int r, int c) { return r * 128 + ((c ^ ((r & 3) << 2)) << 3); }

__global__ __launch_bounds__(256) void gemm_tn_kernel(const u16* __restrict__ A, int64_t lda,
                                                       const u16* __restrict__ B, int64_t ldb, float* __restrict__ C,
                                                       int64_t ldc, int R, int N, int K, int tiles_k, int rows_per_split) {
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

  // staging: tile = 64 rows x 16 chunks = 1024 chunks per operand, 4 per thread
  int srow[4], scol[4], soff[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int id = tid + 256 * i; srow[i] = id >> 4; scol[i] = id & 15; soff[i] = tn_off(srow[i], scol[i]);
  }
  bf16x8 ra_[4], rb_[4];
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  auto gload = [&](int r0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int r = r0 + srow[i];
      const bool rok = r < r_end;
      // a chunk that starts inside [0,N) / [0,K) is read whole: rows are allocated up to the next multiple
      // of 8 columns (checked on the host); outputs past N / K are discarded at the store
      const int ca = n0 + scol[i] * 8, cb = k0 + scol[i] * 8;
      ra_[i] = (rok && ca < N) ? *reinterpret_cast<const bf16x8*>(A + (int64_t)r * lda + ca) : zero;
      rb_[i] = (rok && cb < K) ? *reinterpret_cast<const bf16x8*>(B + (int64_t)r * ldb + cb) : zero;
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      *reinterpret_cast<bf16x8*>(As + buf * BR * 128 + soff[i]) = ra_[i];
      *reinterpret_cast<bf16x8*>(Bs + buf * BR * 128 + soff[i]) = rb_[i];
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
  // 32-row operand block and k-half lh; lane 4q+p supplies LDS row q, columns 4p..4p+3 of the 4x16 block
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int nsteps = (r_end - r_begin + BR - 1) / BR;
  gload(r_begin);
  swrite(0);
  __syncthreads();
  for (int st = 0; st < nsteps; st++) {
    const int cur = st & 1;
    if (st + 1 < nsteps) gload(r_begin + (st + 1) * BR);
    const u16* as = As + cur * BR * 128;
    const u16* bs = Bs + cur * BR * 128;
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {          // 16 reduction rows per MFMA
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int t = 0; t < 2; t++) {
          const int r = 16 * ks + 8 * lh + 4 * t + tq;
          const int ca = wn * 64 + i * 32 + 16 * tg + 4 * tp;      // element column inside the 128-wide tile
          const int cb = wk * 64 + i * 32 + 16 * tg + 4 * tp;
          const bf16x4 va = lds_read_tr16(as + tn_off(r, ca >> 3) + (ca & 7));
          const bf16x4 vb = lds_read_tr16(bs + tn_off(r, cb >> 3) + (cb & 7));
#pragma unroll
          for (int e = 0; e < 4; e++) { af[i][4 * t + e] = va[e]; bfr[i][4 * t + e] = vb[e]; }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (st + 1 < nsteps) swrite(cur ^ 1);
    __syncthreads();
  }
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
        if (n < N) atomicAdd(C + (int64_t)n * ldc + k, acc[i][j][r]);
      }
  }
}

extern "C" int mca_gemm_tn_acc(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                               int64_t R, int64_t N, int64_t K, mca_stream_t stream) {
  if (!A || !B || !C || R <= 0 || N <= 0 || K <= 0) return MCA_E_BADARG;
  if (lda % 8 || ldb % 8 || (uintptr_t)A % 16 || (uintptr_t)B % 16) return MCA_E_ALIGN;
  if (lda < (N + 7) / 8 * 8 || ldb < (K + 7) / 8 * 8 || ldc < K) return MCA_E_BADARG;
  if (R > (1LL << 30)) return MCA_E_UNSUPPORTED;
  const int tiles_n = (int)((N + 127) / 128), tiles_k = (int)((K + 127) / 128);
  const int tiles = tiles_n * tiles_k;
  // split the reduction so that ~4 workgroups per CU are in flight, at least 4 steps of 64 rows each
  int64_t splits = (1024 + tiles - 1) / tiles;
  const int64_t max_splits = (R + 4 * BR - 1) / (4 * BR);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  int64_t rps = (R + splits - 1) / splits;
  rps = (rps + BR - 1) / BR * BR;
  splits = (R + rps - 1) / rps;
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, (unsigned)splits), dim3(256), 0, as_stream(stream), A, lda, B, ldb, C,
                     ldc, (int)R, (int)N, (int)K, tiles_k, (int)rps);
  return launch_status();
}

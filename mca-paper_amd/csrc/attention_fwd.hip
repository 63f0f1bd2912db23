// Block-masked fused attention, forward (reference: model.py:73-105 — q·kᵀ, two masked_fill(-finfo.max),
// softmax, ·v — as called by MCALayer.forward :119 and by the attentive pooling :472-473).
//
// The reference materialises (b,h,N,N) scores and masks.  Here one workgroup owns a 128-row query tile of
// one (sample, head): 4 wavefronts x 32 query rows.  It walks only the 64-key tiles the static fusion
// structure allows for that query tile (CSR list from the host) and skips tiles whose
// keys are all padded in this sample.  Per tile and wavefront:
//     S^T[key][q]  = K · Q^T          A = K rows from LDS (ds_read_b128, XOR-swizzled image),
//                                     B = Q fragments held in registers for the whole kernel
//     online softmax in registers     lane = query (its row max/sum need one cross-half exchange only)
//     O^T[d][q]   += V^T · P^T        A = V^T via ds_read_b64_tr_b16 from the row-major V tile,
//                                     B = the S^T accumulators converted to bf16 in place (no LDS trip)
// with v_mfma_f32_32x32x16_bf16.  Element-wise masking (allowed = (qmask[q] >> keyinfo[key]) & 1) runs
// only on tiles that straddle a structure boundary or contain padded keys.
//
// Semantics kept from the reference: a query row with no allowed, un-padded key gets a UNIFORM
// distribution over ALL nk keys (softmax of a constant row), i.e. its output is the mean of V; such rows
// are marked with lse = +inf for the backward.
#include "common.h"

#define AQ 128      // query rows per workgroup
#define AK 64       // keys per tile
#define DH 64       // head dim (fixed)
#define MAX_KTILES 512
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

MCA_TRACE_BUFFER(attn_fwd)      // knob 8 = 8: s_memtime stamps of one wavefront (tools/trace_attn_fwd.py)
#ifdef MCA_TRACE_BUILD      // the stamps pin the instruction order, so the production build carries none (build.py: trace=True)
#define FW_STAMP() do { if (tracing && ti < 1017) mca_trace_attn_fwd[ti++] = __builtin_readcyclecounter(); } while (0)
#else
#define FW_STAMP() do { } while (0)
#endif

// K tile image: [64 keys][64 d] bf16, 128-byte rows, chunk c (16 B) of row r at c ^ ((r>>1)&7)
__device__ __forceinline__ int k_off(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 7)) << 3); }
// V tile image: same shape, chunk c of row r at c ^ (4*((r>>1)&1)): the 4 rows of a transposed 4x16 read
// fall into 4 distinct 64-byte bank quarters
__device__ __forceinline__ int v_off(int r, int c) { return r * 64 + ((c ^ (((r >> 1) & 1) << 2)) << 3); }

__global__ __launch_bounds__(256) void attn_fwd_kernel(mca_attn_fwd_args a, int dbg) {
  constexpr bool PRESCALED = true;          // q arrives pre-scaled by scale * log2 e (MCA_ATTN_Q_PRESCALED, required)
  __shared__ __attribute__((aligned(16))) u16 lds[2 * 2 * AK * DH];   // K,V double-buffered: 32 KiB
  __shared__ __attribute__((aligned(16))) uint8_t kinfo[2][AK];
  __shared__ __attribute__((aligned(16))) u16 hot_s[2][AK * 16];   // one-hot key groups of the tile (mask product operand)
  __shared__ uint8_t flags_s[MAX_KTILES];       // this sample's key-tile flags
  // This query tile's list entries whose key tile has at least one valid key in this sample, compacted once: reading the
  // CSR entry and the flag of the NEXT live tile at the top of every iteration was a chain of a scalar global load and an
  // LDS read in front of the K / V loads (most of the 1,500 cycles the traced "S" phase took).
  __shared__ uint32_t live_s[MAX_KTILES];
  __shared__ int n_live_s;
  u16* Ks = lds;
  u16* Vs = lds + 2 * AK * DH;

  // XCD-aware order: the query tiles of one (sample, head) read the same K / V, so they are given to one XCD (in launch
  // order they were dealt round all eight and every L2 fetched every K / V: 1,128 MB per launch against 250 MB)
  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));          // knob 9 = 16: launch order (A/B)
  const int qt = a.q_order[lin % (int)gridDim.x];
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int q0 = qt * AQ + wave * 32;
  int qrow = q0 + l31;
  const bool qvalid = qrow < a.nq;
  if (qrow > a.nq - 1) qrow = a.nq - 1;

  // Q fragments (B operand): lane holds Q[q][16s + 8*lh + j]
  bf16x8 qf[4];
  {
    const u16* qp = a.q + (int64_t)b * a.q_bstride + (int64_t)qrow * a.q_ld + h * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  const uint32_t qm = a.qmask[qrow];
  // the mask as a matrix product (mca_build_keyhot): Qblk[q][g] = group g visible ? 0 : -32768 (bf16 0xC700), slot 15 (padded
  // keys) always -32768; this lane holds groups 8 lh .. 8 lh + 7 of its query
  const bool use_hot = a.khot != nullptr;
  bf16x8 qblk;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int g = 8 * lh + j;
    qblk[j] = (g < 15 && ((qm >> g) & 1u)) ? (short)0 : (short)0xC700;
  }
  // scores -> log2 domain; q pre-scaled by scale * log2(e) (MCA_ATTN_Q_PRESCALED): they already are
  const float c2 = PRESCALED ? 1.f : a.scale * 1.4426950408889634f;

  f32x16 o[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[n][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  const uint8_t* kinfo_g = a.keyinfo + (int64_t)b * a.nk_pad;
  const u16* khot_g = use_hot ? a.khot + (int64_t)b * a.nk_pad * 16 : nullptr;
  {
    const uint8_t* flags_g = a.ktile_flags + (int64_t)b * a.n_ktiles;
    for (int i = tid; i < a.n_ktiles; i += 256) flags_s[i] = flags_g[i];
  }
  __syncthreads();

  // staging: 512 chunks of 16 B per tile per operand, 2 per thread.  The per-lane element offset inside a tile is
  // loop-invariant; the tile base is wave-uniform (scalar), so the loads use the SGPR-base + VGPR-offset form and no
  // address register is recycled while the prefetch is in flight.  Only the last key tile (which may run past nk)
  // takes the per-lane clamped path.
  int srow[2], sc[2];
  unsigned loff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int id = tid + 256 * i; srow[i] = id >> 3; sc[i] = id & 7;
    loff[i] = (unsigned)(srow[i] * (int)a.kv_ld + sc[i] * 8);
  }
  const int last_kt = a.n_ktiles - 1;
  bf16x8 rk[2], rv[2];
  uint32_t rinfo = 0;
  uint4 rhot = make_uint4(0, 0, 0, 0);
  auto gload = [&](int kt) {
    if (use_hot) { if (tid < 128) rhot = *reinterpret_cast<const uint4*>(khot_g + ((int64_t)kt * AK + (tid >> 1)) * 16 + (tid & 1) * 8); }
    else if (tid < 16) rinfo = *reinterpret_cast<const uint32_t*>(kinfo_g + kt * AK + tid * 4);
    const u16* kb = kbase + (int64_t)kt * AK * a.kv_ld;
    const u16* vb = vbase + (int64_t)kt * AK * a.kv_ld;
    if (kt != last_kt) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        rk[i] = *reinterpret_cast<const bf16x8*>(kb + loff[i]);
        rv[i] = *reinterpret_cast<const bf16x8*>(vb + loff[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int key = kt * AK + srow[i]; if (key > a.nk - 1) key = a.nk - 1;
        rk[i] = *reinterpret_cast<const bf16x8*>(kbase + (int64_t)key * a.kv_ld + sc[i] * 8);
        rv[i] = *reinterpret_cast<const bf16x8*>(vbase + (int64_t)key * a.kv_ld + sc[i] * 8);
      }
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      *reinterpret_cast<bf16x8*>(Ks + buf * AK * DH + k_off(srow[i], sc[i])) = rk[i];
      *reinterpret_cast<bf16x8*>(Vs + buf * AK * DH + v_off(srow[i], sc[i])) = rv[i];
    }
    if (use_hot) { if (tid < 128) *reinterpret_cast<uint4*>(&hot_s[buf][(tid >> 1) * 16 + (tid & 1) * 8]) = rhot; }
    else if (tid < 16) *reinterpret_cast<uint32_t*>(&kinfo[buf][tid * 4]) = rinfo;
  };

  // the tile list of this query tile, minus tiles whose keys are all padded in this sample (wavefront 0 compacts it)
  if (wave == 0) {
    const int lb = a.q_ptr[qt], le = a.q_ptr[qt + 1];
    int n = 0;
    for (int i0 = lb; i0 < le; i0 += 64) {
      const int i = i0 + lane;
      const uint32_t e = i < le ? a.q_kt[i] : 0u;
      const bool keep = i < le && flags_s[e & 0x7fffffffu] != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) live_s[n + __popcll(m & ((1ull << lane) - 1ull))] = e;
      n += __popcll(m);
    }
    if (lane == 0) n_live_s = n;
  }
  __syncthreads();
  const int it_end = n_live_s;
  int it = 0;
  int buf = 0;
  if (it < it_end) { gload((int)(live_s[0] & 0x7fffffffu)); swrite(0); }
  __syncthreads();

  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const bool tracing = (dbg & 8) && lin0 == 8 * 3 && tid == 0;
  int ti = 0;
  const uint64_t rt0 = tracing ? __builtin_amdgcn_s_memrealtime() : 0, cy0 = tracing ? __builtin_readcyclecounter() : 0;
  while (it < it_end) {
    FW_STAMP();
    const uint32_t ent = live_s[it];
    const int kt = (int)(ent & 0x7fffffffu);
    const bool need_mask = (ent >> 31) == 0 || (flags_s[kt] != 2);
    const int nit = it + 1;
    if (nit < it_end) gload((int)(live_s[nit] & 0x7fffffffu));

    const u16* ks = Ks + buf * AK * DH;
    const u16* vs = Vs + buf * AK * DH;
    // ---- S^T = K Q^T : two 32-key blocks
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) s[kb][r] = 0.f;
#pragma unroll
      for (int st = 0; st < 4; st++) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + k_off(kb * 32 + l31, 2 * st + lh));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kb], 0, 0, 0);
      }
    }
    // ---- masking (structure boundary / padded keys / keys past nk) as one more product into the same accumulators
    if (need_mask && use_hot) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(&hot_s[buf][(kb * 32 + l31) * 16 + 8 * lh]);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, qblk, s[kb], 0, 0, 0);
      }
    }
    FW_STAMP();
    // ---- the same, element by element (no one-hot operand: more than 15 key groups)
    if (need_mask && !use_hot) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const uint32_t info4 = *reinterpret_cast<const uint32_t*>(&kinfo[buf][kb * 32 + 8 * g + 4 * lh]);
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const uint32_t grp = (info4 >> (8 * e)) & 0xffu;
            const bool ok = (qm >> grp) & 1u;
            s[kb][4 * g + e] = ok ? s[kb][4 * g + e] : -INFINITY;
          }
        }
    }
    FW_STAMP();
    // ---- online softmax (log2 domain), lane = query
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
    const float m_new = fmaxf(m_run, PRESCALED ? mx : mx * c2);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float rs = 0.f;
    bf16x8 pb[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float p = __builtin_amdgcn_exp2f(PRESCALED ? s[kb][8 * sp + j] - m_use : fmaf(s[kb][8 * sp + j], c2, -m_use));
          rs += p;
          pb[kb][sp][j] = (short)f2bf(p);
        }
    rs += __shfl_xor(rs, 32, WAVE);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[n][r] *= alpha;
    // ---- O^T += V^T P^T : A = V^T (transposed LDS reads), B = P^T (accumulators as operand; element j of
    //      k-step sp carries key 16sp + 8(j>>2) + 4lh + (j&3) of the 32-key block)
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
          bf16x8 vf;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int key = kb * 32 + 16 * sp + 8 * t + 4 * lh + tq;
            const int d = n * 32 + 16 * tg + 4 * tp;
            const bf16x4 v4 = lds_read_tr16(vs + v_off(key, d >> 3) + (d & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) vf[4 * t + e] = v4[e];
          }
          o[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kb][sp], o[n], 0, 0, 0);
        }

    FW_STAMP();
    if (nit < it_end) swrite(buf ^ 1);
    FW_STAMP();
    __syncthreads();
    FW_STAMP();
    buf ^= 1;
    it = nit;
  }

#ifdef MCA_TRACE_BUILD
  if (tracing) {      // shader clock of this launch: cycles per 100 MHz tick
    mca_trace_attn_fwd[1020] = __builtin_amdgcn_s_memrealtime() - rt0; mca_trace_attn_fwd[1021] = __builtin_readcyclecounter() - cy0;
  }
#endif
  // ---- epilogue (a row that met blocked keys only: no tile at all, or every score carried the -32768 of the mask product)
  const bool uniform = !(l_run > 0.f) || m_run < -16384.f * c2;
  const float inv = uniform ? 0.f : 1.f / l_run;
  if (qvalid) {
    if (lh == 0) a.lse[((int64_t)b * a.heads + h) * a.nq + qrow] = uniform ? INFINITY : m_run + log2f(l_run);
    u16* op = a.o + (int64_t)b * a.o_bstride + (int64_t)qrow * a.o_ld + h * DH;
    const float* vm = a.vmean + (int64_t)b * a.heads * DH + h * DH;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int d = n * 32 + 8 * g + 4 * lh;
        float v0, v1, v2, v3;
        if (uniform) { v0 = vm[d]; v1 = vm[d + 1]; v2 = vm[d + 2]; v3 = vm[d + 3]; }
        else { v0 = o[n][4 * g] * inv; v1 = o[n][4 * g + 1] * inv; v2 = o[n][4 * g + 2] * inv; v3 = o[n][4 * g + 3] * inv; }
        uint2 pk; pk.x = pack2bf(v0, v1); pk.y = pack2bf(v2, v3);
        *reinterpret_cast<uint2*>(op + d) = pk;
      }
  }
}



// =====================================================================================================
// Forward, LDS-DMA staging (production when the mask product's one-hot operand exists): the arithmetic of attn_fwd_kernel
// with the K / V / one-hot tiles moved global -> LDS by global_load_lds_dwordx4: no staging registers, no ds_write pass, the
// XOR swizzles applied on the SOURCE address (the DMA writes lane-linear).  126 registers instead of 162: FOUR workgroups per
// CU instead of three (39.4 KB of LDS each).  Measured against attn_fwd_kernel on one box (tools/bench_attn.py): b = 32
// unpadded 433 -> 420 us, padded 254 -> 238 us, b = 8 127 -> 109 us; the fourth wavefront per SIMD buys 3 % where the
// chip is full and 14 % where it is not: the loop is bound by vector / matrix ISSUE per SIMD, not by latency.
// Same arithmetic in the same order as attn_fwd_kernel: the two give the same bits (tests/test_attention_gpu.py).
// =====================================================================================================
__global__ __launch_bounds__(256, 4) void attn_fwd4_kernel(mca_attn_fwd_args a, int dbg) {
  __shared__ __attribute__((aligned(16))) u16 lds[2 * 2 * AK * DH + 2 * AK * 16];   // K, V double-buffered (32 KiB) + one-hot tiles (4 KiB)
  __shared__ uint8_t flags_s[MAX_KTILES];
  __shared__ uint32_t live_s[MAX_KTILES];
  __shared__ int n_live_s;
  u16* Ks = lds;
  u16* Vs = lds + 2 * AK * DH;
  u16* Hs = lds + 4 * AK * DH;

  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));
  const int qt = a.q_order[lin % (int)gridDim.x];
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int q0 = qt * AQ + wave * 32;
  int qrow = q0 + l31;
  const bool qvalid = qrow < a.nq;
  if (qrow > a.nq - 1) qrow = a.nq - 1;

  bf16x8 qf[4];
  {
    const u16* qp = a.q + (int64_t)b * a.q_bstride + (int64_t)qrow * a.q_ld + h * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  const uint32_t qm = a.qmask[qrow];
  const uint32_t qm8 = ((qm >> (8 * lh)) & 0xffu) & (lh ? 0x7fu : 0xffu);

  f32x16 o[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[n][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  const u16* khot_g = a.khot + (int64_t)b * a.nk_pad * 16;
  {
    const uint8_t* flags_g = a.ktile_flags + (int64_t)b * a.n_ktiles;
    for (int i = tid; i < a.n_ktiles; i += 256) flags_s[i] = flags_g[i];
  }
  __syncthreads();

  // DMA pieces: piece p = 256 i + tid of a tile image lands at byte 16 p; it must hold chunk c = slot ^ swizzle(row) of row
  // p >> 3 (slot = p & 7): the swizzle sits on the SOURCE column
  int prow[2];
  unsigned koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = 256 * i + tid, r = p >> 3, sl = p & 7;
    prow[i] = r;
    koff[i] = (unsigned)(r * (int)a.kv_ld + ((sl ^ ((r >> 1) & 7)) << 3));
    voff[i] = (unsigned)(r * (int)a.kv_ld + ((sl ^ (((r >> 1) & 1) << 2)) << 3));
  }
  const int last_kt = a.n_ktiles - 1;
  auto stage = [&](int kt, int buf) {
    const u16* kb = kbase + (int64_t)kt * AK * a.kv_ld;
    const u16* vb = vbase + (int64_t)kt * AK * a.kv_ld;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      unsigned ko = koff[i], vo = voff[i];
      if (kt == last_kt) {          // rows past nk: re-read the last valid row (masked by its one-hot row)
        int key = kt * AK + prow[i]; const int over = key - (a.nk - 1);
        if (over > 0) { ko -= (unsigned)(over * (int)a.kv_ld); vo -= (unsigned)(over * (int)a.kv_ld); }
      }
      u16* dk = Ks + buf * AK * DH + (i * 4 + wave) * 512;
      u16* dv = Vs + buf * AK * DH + (i * 4 + wave) * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + ko), (__attribute__((address_space(3))) void*)dk, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + vo), (__attribute__((address_space(3))) void*)dv, 16, 0, 0);
    }
    if (wave < 2) {                 // 64 keys x 16 one-hot columns = 128 pieces
      u16* dh = Hs + buf * AK * 16 + wave * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(khot_g + (int64_t)kt * (AK * 16) + tid * 8),
                                       (__attribute__((address_space(3))) void*)dh, 16, 0, 0);
    }
  };

  if (wave == 0) {
    const int lb = a.q_ptr[qt], le = a.q_ptr[qt + 1];
    int n = 0;
    for (int i0 = lb; i0 < le; i0 += 64) {
      const int i = i0 + lane;
      const uint32_t e = i < le ? a.q_kt[i] : 0u;
      const uint8_t fl = i < le ? flags_s[e & 0x7fffffffu] : (uint8_t)0;
      const bool keep = fl != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) live_s[n + __popcll(m & ((1ull << lane) - 1ull))] = (e & 0x7fffffffu) | (((e >> 31) && fl == 2) ? 0x80000000u : 0u);
      n += __popcll(m);
    }
    if (lane == 0) n_live_s = n;
  }
  __syncthreads();
  const int it_end = n_live_s;
  int buf = 0;
  if (it_end > 0) stage((int)(live_s[0] & 0x7fffffffu), 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  for (int it = 0; it < it_end; it++) {
    const uint32_t ent = live_s[it];
    const bool need_mask = (ent >> 31) == 0;
    if (it + 1 < it_end) stage((int)(live_s[it + 1] & 0x7fffffffu), buf ^ 1);
    const u16* ks = Ks + buf * AK * DH;
    const u16* vs = Vs + buf * AK * DH;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) s[kb][r] = 0.f;
#pragma unroll
      for (int st = 0; st < 4; st++) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + k_off(kb * 32 + l31, 2 * st + lh));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], s[kb], 0, 0, 0);
      }
    }
    if (need_mask) {
      u32x4v qb;
#pragma unroll
      for (int w = 0; w < 4; w++)
        qb[w] = (((qm8 >> (2 * w)) & 1u) ? 0u : 0xC700u) | (((qm8 >> (2 * w + 1)) & 1u) ? 0u : 0xC7000000u);
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(Hs + buf * AK * 16 + (kb * 32 + l31) * 16 + 8 * lh);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, *reinterpret_cast<const bf16x8*>(&qb), s[kb], 0, 0, 0);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
    float rs = 0.f;
    u32x4v pw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float p0 = __builtin_amdgcn_exp2f(s[kb][8 * sp + j] - m_use), p1 = __builtin_amdgcn_exp2f(s[kb][8 * sp + j + 1] - m_use);
          rs += p0; rs += p1;          // (one after the other, as attn_fwd_kernel adds them: the two kernels give the same bits)
          pw[kb][sp][j >> 1] = pack2bf_pk(p0, p1);
        }
    rs += __shfl_xor(rs, 32, WAVE);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[n][r] *= alpha;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
          bf16x8 vf;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int key = kb * 32 + 16 * sp + 8 * t + 4 * lh + tq;
            const int d = n * 32 + 16 * tg + 4 * tp;
            const bf16x4 v4 = lds_read_tr16(vs + v_off(key, d >> 3) + (d & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) vf[4 * t + e] = v4[e];
          }
          o[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, *reinterpret_cast<const bf16x8*>(&pw[kb][sp]), o[n], 0, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of the next tile have landed ...
    __syncthreads();                                          // ... everybody's have, and this tile's reads are done
    buf ^= 1;
  }

  const bool uniform = !(l_run > 0.f) || m_run < -16384.f;
  const float inv = uniform ? 0.f : 1.f / l_run;
  if (qvalid) {
    if (lh == 0) a.lse[((int64_t)b * a.heads + h) * a.nq + qrow] = uniform ? INFINITY : m_run + log2f(l_run);
    u16* op = a.o + (int64_t)b * a.o_bstride + (int64_t)qrow * a.o_ld + h * DH;
    const float* vm = a.vmean + (int64_t)b * a.heads * DH + h * DH;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int d = n * 32 + 8 * g + 4 * lh;
        float v0, v1, v2, v3;
        if (uniform) { v0 = vm[d]; v1 = vm[d + 1]; v2 = vm[d + 2]; v3 = vm[d + 3]; }
        else { v0 = o[n][4 * g] * inv; v1 = o[n][4 * g + 1] * inv; v2 = o[n][4 * g + 2] * inv; v3 = o[n][4 * g + 3] * inv; }
        uint2 pk; pk.x = pack2bf(v0, v1); pk.y = pack2bf(v2, v3);
        *reinterpret_cast<uint2*>(op + d) = pk;
      }
  }
}

// =====================================================================================================
// The LDS-DMA kernel with a LAZY softmax reference (MCA_ATTN_LAZY_REFERENCE; round 4): same tiling, staging and mask product as
// attn_fwd4_kernel, but the score accumulators start from -m (the MFMA C operand) and m moves only in a rare wave-uniform slow
// path.  Per 32 x 64 block that removes 32 subtractions, the 32 multiplies that rescale
// O, the exponential of the rescale factor and both cross-lane exchanges of the online softmax: a third of the vector
// instructions of a loop that is bound by vector issue.  16 registers more (-m): three wavefronts per SIMD instead of four,
// which this kernel does not notice (round 3: 338.5 us with three against 338.6 with four).  Results differ from
// attn_fwd4_kernel by rounding only (P is rounded to bf16 against another reference); knob 15 = t + 1 sets the threshold t
// (1: the reference follows every increase, the textbook recurrence).  NOT the default: the re-drawn rounding pattern moves the
// cancellation-heavy gradient statistics of the CMU golden step (three bias-gradient norms 1.0 % -> 4.4 % off the reference's
// numbers, pooled embeddings 6.1e-4 either way), and parity margins outrank 0.75 % of the step.
// =====================================================================================================
__global__ __launch_bounds__(256, 3) void attn_fwd4l_kernel(mca_attn_fwd_args a, int dbg, float thr) {
  __shared__ __attribute__((aligned(16))) u16 lds[2 * 2 * AK * DH + 2 * AK * 16];   // K, V double-buffered (32 KiB) + one-hot tiles (4 KiB)
  __shared__ uint8_t flags_s[MAX_KTILES];
  __shared__ uint32_t live_s[MAX_KTILES];
  __shared__ int n_live_s;
  u16* Ks = lds;
  u16* Vs = lds + 2 * AK * DH;
  u16* Hs = lds + 4 * AK * DH;

  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));
  const int qt = a.q_order[lin % (int)gridDim.x];
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int q0 = qt * AQ + wave * 32;
  int qrow = q0 + l31;
  const bool qvalid = qrow < a.nq;
  if (qrow > a.nq - 1) qrow = a.nq - 1;

  bf16x8 qf[4];
  {
    const u16* qp = a.q + (int64_t)b * a.q_bstride + (int64_t)qrow * a.q_ld + h * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; s++) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  const uint32_t qm = a.qmask[qrow];
  const uint32_t qm8 = ((qm >> (8 * lh)) & 0xffu) & (lh ? 0x7fu : 0xffu);

  f32x16 o[2];
#pragma unroll
  for (int n = 0; n < 2; n++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[n][r] = 0.f;
  // lazy reference maximum : the S^T accumulators START from -m, so the scores leave the matrix pipe as
  // S - m; m moves only when a score exceeds it by more than thr, or when a row that has accumulated nothing meets its first
  // REAL key (blocked scores, -32768 from the mask product, never set it); l_run is this LANE's share of the row sum
  f32x16 negm;
#pragma unroll
  for (int r = 0; r < 16; r++) negm[r] = 0.f;
  float l_run = 0.f;

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  const u16* khot_g = a.khot + (int64_t)b * a.nk_pad * 16;
  {
    const uint8_t* flags_g = a.ktile_flags + (int64_t)b * a.n_ktiles;
    for (int i = tid; i < a.n_ktiles; i += 256) flags_s[i] = flags_g[i];
  }
  __syncthreads();

  // DMA pieces: piece p = 256 i + tid of a tile image lands at byte 16 p; it must hold chunk c = slot ^ swizzle(row) of row
  // p >> 3 (slot = p & 7): the swizzle sits on the SOURCE column
  int prow[2];
  unsigned koff[2], voff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int p = 256 * i + tid, r = p >> 3, sl = p & 7;
    prow[i] = r;
    koff[i] = (unsigned)(r * (int)a.kv_ld + ((sl ^ ((r >> 1) & 7)) << 3));
    voff[i] = (unsigned)(r * (int)a.kv_ld + ((sl ^ (((r >> 1) & 1) << 2)) << 3));
  }
  const int last_kt = a.n_ktiles - 1;
  auto stage = [&](int kt, int buf) {
    const u16* kb = kbase + (int64_t)kt * AK * a.kv_ld;
    const u16* vb = vbase + (int64_t)kt * AK * a.kv_ld;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      unsigned ko = koff[i], vo = voff[i];
      if (kt == last_kt) {          // rows past nk: re-read the last valid row (masked by its one-hot row)
        int key = kt * AK + prow[i]; const int over = key - (a.nk - 1);
        if (over > 0) { ko -= (unsigned)(over * (int)a.kv_ld); vo -= (unsigned)(over * (int)a.kv_ld); }
      }
      u16* dk = Ks + buf * AK * DH + (i * 4 + wave) * 512;
      u16* dv = Vs + buf * AK * DH + (i * 4 + wave) * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kb + ko), (__attribute__((address_space(3))) void*)dk, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vb + vo), (__attribute__((address_space(3))) void*)dv, 16, 0, 0);
    }
    if (wave < 2) {                 // 64 keys x 16 one-hot columns = 128 pieces
      u16* dh = Hs + buf * AK * 16 + wave * 512;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(khot_g + (int64_t)kt * (AK * 16) + tid * 8),
                                       (__attribute__((address_space(3))) void*)dh, 16, 0, 0);
    }
  };

  if (wave == 0) {
    const int lb = a.q_ptr[qt], le = a.q_ptr[qt + 1];
    int n = 0;
    for (int i0 = lb; i0 < le; i0 += 64) {
      const int i = i0 + lane;
      const uint32_t e = i < le ? a.q_kt[i] : 0u;
      const uint8_t fl = i < le ? flags_s[e & 0x7fffffffu] : (uint8_t)0;
      const bool keep = fl != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) live_s[n + __popcll(m & ((1ull << lane) - 1ull))] = (e & 0x7fffffffu) | (((e >> 31) && fl == 2) ? 0x80000000u : 0u);
      n += __popcll(m);
    }
    if (lane == 0) n_live_s = n;
  }
  __syncthreads();
  const int it_end = n_live_s;
  int buf = 0;
  if (it_end > 0) stage((int)(live_s[0] & 0x7fffffffu), 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  for (int it = 0; it < it_end; it++) {
    const uint32_t ent = live_s[it];
    const bool need_mask = (ent >> 31) == 0;
    if (it + 1 < it_end) stage((int)(live_s[it + 1] & 0x7fffffffu), buf ^ 1);
    const u16* ks = Ks + buf * AK * DH;
    const u16* vs = Vs + buf * AK * DH;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int st = 0; st < 4; st++) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + k_off(kb * 32 + l31, 2 * st + lh));
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[st], st == 0 ? negm : s[kb], 0, 0, 0);
      }
    }
    if (need_mask) {
      u32x4v qb;
#pragma unroll
      for (int w = 0; w < 4; w++)
        qb[w] = (((qm8 >> (2 * w)) & 1u) ? 0u : 0xC700u) | (((qm8 >> (2 * w + 1)) & 1u) ? 0u : 0xC7000000u);
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(Hs + buf * AK * 16 + (kb * 32 + l31) * 16 + 8 * lh);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, *reinterpret_cast<const bf16x8*>(&qb), s[kb], 0, 0, 0);
      }
    }
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[0][r]), s[0][r + 1]);
    mx = fmaxf(fmaxf(mx, s[0][15]), s[1][0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[1][r]), s[1][r + 1]);
    mx = fmaxf(mx, s[1][15]);
    const bool over = (mx > thr) | ((l_run == 0.f) & (mx > -16384.f));
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(over) != 0ull, 0)) {
      // slow path (wave-uniform, rare): everything at the old reference is rescaled exactly once.  Every P.V of earlier tiles
      // is complete (this kernel does not pipeline across tiles), this tile's exponentials come afterwards.
      const float mfull = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
      const float lrow = l_run + __shfl_xor(l_run, 32, WAVE);
      const bool empty = !(lrow > 0.f), real = mfull > -16384.f;
      const float delta = (empty && real) ? mfull : (mfull > thr ? mfull : 0.f);
      const float alpha = empty ? 0.f : __builtin_amdgcn_exp2f(-delta);
      l_run *= alpha;
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[n][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; r++) { negm[r] -= delta; s[0][r] -= delta; s[1][r] -= delta; }
    }
    float rs = 0.f;
    u32x4v pw[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float p0 = __builtin_amdgcn_exp2f(s[kb][8 * sp + j]), p1 = __builtin_amdgcn_exp2f(s[kb][8 * sp + j + 1]);
          rs += p0; rs += p1;
          pw[kb][sp][j >> 1] = pack2bf_pk(p0, p1);
        }
    l_run += rs;
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int n = 0; n < 2; n++) {
          bf16x8 vf;
#pragma unroll
          for (int t = 0; t < 2; t++) {
            const int key = kb * 32 + 16 * sp + 8 * t + 4 * lh + tq;
            const int d = n * 32 + 16 * tg + 4 * tp;
            const bf16x4 v4 = lds_read_tr16(vs + v_off(key, d >> 3) + (d & 7));
#pragma unroll
            for (int e = 0; e < 4; e++) vf[4 * t + e] = v4[e];
          }
          o[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, *reinterpret_cast<const bf16x8*>(&pw[kb][sp]), o[n], 0, 0, 0);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's pieces of the next tile have landed ...
    __syncthreads();                                          // ... everybody's have, and this tile's reads are done
    buf ^= 1;
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, WAVE);
  const float m_run = -negm[0];
  const bool uniform = !(l_tot > 0.f);          // only real keys ever contribute: a row of blocked keys sums exact zeros
  const float inv = uniform ? 0.f : 1.f / l_tot;
  if (qvalid) {
    if (lh == 0) a.lse[((int64_t)b * a.heads + h) * a.nq + qrow] = uniform ? INFINITY : m_run + log2f(l_tot);
    u16* op = a.o + (int64_t)b * a.o_bstride + (int64_t)qrow * a.o_ld + h * DH;
    const float* vm = a.vmean + (int64_t)b * a.heads * DH + h * DH;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int d = n * 32 + 8 * g + 4 * lh;
        float v0, v1, v2, v3;
        if (uniform) { v0 = vm[d]; v1 = vm[d + 1]; v2 = vm[d + 2]; v3 = vm[d + 3]; }
        else { v0 = o[n][4 * g] * inv; v1 = o[n][4 * g + 1] * inv; v2 = o[n][4 * g + 2] * inv; v3 = o[n][4 * g + 3] * inv; }
        uint2 pk; pk.x = pack2bf(v0, v1); pk.y = pack2bf(v2, v3);
        *reinterpret_cast<uint2*>(op + d) = pk;
      }
  }
}

extern "C" int mca_attn_fwd(const mca_attn_fwd_args* a, mca_stream_t stream) {
  if (!a || !a->q || !a->k || !a->v || !a->o || !a->lse || !a->qmask || !a->keyinfo || !a->ktile_flags || !a->q_ptr ||
      !a->q_kt || !a->q_order || !a->vmean)
    return MCA_E_BADARG;
  if (a->batch <= 0 || a->heads <= 0 || a->nq <= 0 || a->nk <= 0) return MCA_E_BADARG;
  if (a->n_qtiles != (a->nq + AQ - 1) / AQ || a->n_ktiles != (a->nk + AK - 1) / AK) return MCA_E_BADARG;
  if (a->nk_pad < a->n_ktiles * AK || a->nk_pad % 4) return MCA_E_BADARG;
  if (a->q_ld % 8 || a->kv_ld % 8 || a->o_ld % 4 || a->q_bstride % 8 || a->kv_bstride % 8 || a->o_bstride % 4) return MCA_E_ALIGN;
  if ((uintptr_t)a->q % 16 || (uintptr_t)a->k % 16 || (uintptr_t)a->v % 16 || (uintptr_t)a->o % 8 ||
      (uintptr_t)a->keyinfo % 4)
    return MCA_E_ALIGN;
  if (a->heads > 65535 || a->batch > 65535 || a->n_ktiles > MAX_KTILES) return MCA_E_UNSUPPORTED;
  const dim3 grid(a->n_qtiles, a->heads, a->batch);
  if (!(a->flags & MCA_ATTN_Q_PRESCALED)) return MCA_E_UNSUPPORTED;          // (the un-prescaled forms left the library in round 3)
  // production: the LDS-DMA kernels (they need the mask product's one-hot operand and 32-bit tile offsets) - with the lazy softmax
  // reference when the caller asks for it (the engine's default since round 5: profiles/r05_lazy_softmax_seed_study.txt); structures
  // with more than 15 key groups (no khot) take the register-staged kernel with the element-wise mask
  if ((uintptr_t)a->khot % 16) return MCA_E_ALIGN;
  const bool dma = a->khot && (int64_t)a->kv_ld * 64 < (1ll << 30);
  if (dma && (a->flags & MCA_ATTN_LAZY_REFERENCE))
    hipLaunchKernelGGL(attn_fwd4l_kernel, grid, dim3(256), 0, as_stream(stream), *a, mca_knobs[9], 12.f);
  else if (dma)
    hipLaunchKernelGGL(attn_fwd4_kernel, grid, dim3(256), 0, as_stream(stream), *a, mca_knobs[9]);
  else
    hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), 0, as_stream(stream), *a, mca_knobs[9] | mca_knobs[8]);
  return launch_status();
}

// =====================================================================================================
// keyinfo / tile flags, vmean
// =====================================================================================================
__global__ __launch_bounds__(64) void build_keyinfo_kernel(const uint8_t* __restrict__ padding,
                                                            const uint8_t* __restrict__ kgroup,
                                                            uint8_t* __restrict__ keyinfo, uint8_t* __restrict__ flags,
                                                            int nk, int nk_pad, int n_ktiles) {
  // one wavefront per (sample, 64-key tile); tiles past n_ktiles only fill the pad region
  const int b = blockIdx.y, kt = blockIdx.x, lane = threadIdx.x;
  const int key = kt * AK + lane;
  uint8_t info = 31;
  if (key < nk && !padding[(int64_t)b * nk + key]) info = kgroup[key];
  if (key < nk_pad) keyinfo[(int64_t)b * nk_pad + key] = info;
  const unsigned long long valid = __ballot(info != 31);
  if (lane == 0 && kt < n_ktiles) {
    const int in_tile = (nk - kt * AK) < AK ? (nk - kt * AK) : AK;
    const int nvalid = __popcll(valid);
    flags[(int64_t)b * n_ktiles + kt] = nvalid == 0 ? 0 : (nvalid == AK && in_tile == AK ? 2 : 1);
  }
}
extern "C" int mca_build_keyinfo(const uint8_t* padding, const uint8_t* kgroup, uint8_t* keyinfo,
                                 uint8_t* ktile_flags, int batch, int nk, int nk_pad, mca_stream_t stream) {
  if (!padding || !kgroup || !keyinfo || !ktile_flags || batch <= 0 || nk <= 0) return MCA_E_BADARG;
  const int n_ktiles = (nk + AK - 1) / AK;
  if (nk_pad < n_ktiles * AK) return MCA_E_BADARG;
  const int ntile_pad = (nk_pad + AK - 1) / AK;
  hipLaunchKernelGGL(build_keyinfo_kernel, dim3(ntile_pad, batch), dim3(64), 0, as_stream(stream), padding, kgroup,
                     keyinfo, ktile_flags, nk, nk_pad, n_ktiles);
  return launch_status();
}

// khot[b, j, 0..15] = one-hot (bf16 1.0) of min(keyinfo[b, j], 15): padded keys and keys past nk (31) land in slot 15
__global__ __launch_bounds__(256) void build_keyhot_kernel(const uint8_t* __restrict__ keyinfo, u16* __restrict__ khot, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int g = keyinfo[i] >= 15 ? 15 : keyinfo[i];
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = (g >> 1) == k ? (0x3F80u << (16 * (g & 1))) : 0u;
  uint4* dst = reinterpret_cast<uint4*>(khot + i * 16);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
extern "C" int mca_build_keyhot(const uint8_t* keyinfo, uint16_t* khot, int batch, int nk_pad, mca_stream_t stream) {
  if (!keyinfo || !khot || batch <= 0 || nk_pad <= 0) return MCA_E_BADARG;
  if ((uintptr_t)khot % 16) return MCA_E_ALIGN;
  const int64_t n = (int64_t)batch * nk_pad;
  hipLaunchKernelGGL(build_keyhot_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), keyinfo, khot, n);
  return launch_status();
}

// vmean[b, c] = (1/nk) sum_j V[b, j, c]   c in [0, heads*64).  grid (64-column chunks, b), 512 threads: 8 lanes cover the 64
// columns of a row (16 bytes each), 64 row groups stride the keys; the partial sums meet in LDS in a FIXED order, so the result
// is bitwise reproducible (it is the output of every query row without a valid key: an atomic here made the whole step's
// gradients differ from run to run at the 1e-3 level).
__global__ __launch_bounds__(512) void vmean_kernel(const u16* __restrict__ V, int64_t bstride, int64_t ld,
                                                     float* __restrict__ vmean, int nk, int cols,
                                                     const int32_t* __restrict__ present, int32_t full_bits) {
  __shared__ float red[64][65];
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  if (present && present[b] == full_bits) return;          // every modality present: no row of this sample is fully masked
  const int cl = threadIdx.x & 7, rg = threadIdx.x >> 3;
  const u16* base = V + (int64_t)b * bstride + c0 + cl * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = rg; j < nk; j += 64) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + (int64_t)j * ld);
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] += bf2f((u16)v[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; e++) red[rg][cl * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int r = 0; r < 64; r++) t += red[r][threadIdx.x];
    vmean[(int64_t)b * cols + c0 + threadIdx.x] = t * (1.f / (float)nk);
  }
}
extern "C" int mca_attn_vmean(const uint16_t* V, int64_t kv_bstride, int64_t kv_ld, float* vmean, int batch, int nk,
                              int heads, mca_stream_t stream) {
  if (!V || !vmean || batch <= 0 || nk <= 0 || heads <= 0) return MCA_E_BADARG;
  const int cols = heads * DH;
  if (kv_ld % 8 || kv_bstride % 8 || (uintptr_t)V % 16) return MCA_E_ALIGN;
  if (batch > 65535) return MCA_E_UNSUPPORTED;
  hipLaunchKernelGGL(vmean_kernel, dim3(heads, batch), dim3(512), 0, as_stream(stream), V, kv_bstride, kv_ld, vmean, nk, cols,
                     (const int32_t*)nullptr, 0);
  return launch_status();
}
extern "C" int mca_attn_vmean_if_needed(const uint16_t* V, int64_t kv_bstride, int64_t kv_ld, float* vmean, int batch, int nk,
                                        int heads, const int32_t* present, int32_t full_bits, mca_stream_t stream) {
  if (!V || !vmean || !present || batch <= 0 || nk <= 0 || heads <= 0) return MCA_E_BADARG;
  const int cols = heads * DH;
  if (kv_ld % 8 || kv_bstride % 8 || (uintptr_t)V % 16) return MCA_E_ALIGN;
  if (batch > 65535) return MCA_E_UNSUPPORTED;
  hipLaunchKernelGGL(vmean_kernel, dim3(heads, batch), dim3(512), 0, as_stream(stream), V, kv_bstride, kv_ld, vmean, nk, cols,
                     present, full_bits);
  return launch_status();
}

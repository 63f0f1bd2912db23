// Block-masked fused attention, forward, round-4 structure (reference: model.py:87-99 — q·kᵀ, two masked_fill(-finfo.max),
// softmax, ·v — as called by MCALayer.forward :119).  Same semantics and operand formats as attention_fwd.hip (which keeps the
// pooling attention and the structures with more than 15 key groups); what changes is the shape of the work:
//
//   * a workgroup = 4 wavefronts = one QUERY BLOCK of up to 256 rows cut along the fusion structure by the host
//     (structure.block_schedule: a block never straddles two modalities unless both are small), ONE wavefront per SIMD,
//     64 query rows per wavefront (two 32-row blocks) on the whole 512-entry register file: scores, P, O^T and -m in the
//     vector half, every operand only the matrix pipe reads (Q, K / V^T / one-hot fragments, mask words) in the accumulator
//     half; every K and V^T fragment read from LDS feeds TWO matrix instructions (32 rows per wavefront: one);
//   * K / V / one-hot tiles arrive by LDS-DMA into a ring of four stages; ONE raw s_barrier per tile, placed between the two
//     steps of a tile: behind it the next tile is readable (its fragments are prefetched during the second step) and the
//     stage of the tile two ahead is free (its DMA pieces are issued one at a time in the second step's slots);
//   * every fragment read of a step is issued during the step before it, one per slot: no LDS latency at a step's start;
//   * the running maximum is LAZY: the accumulators of S^T start from -m (the MFMA C operand), so the scores come out of the
//     matrix pipe as S - m and exp2 applies to them directly: no subtraction, no rescale of O, no cross-lane exchange per
//     tile.  Only when a new score exceeds m by more than F64_THR (or a row meets its first real key) does a wave-uniform slow
//     path move m (rescaling O, l and the pending scores exactly once);
//   * software pipeline over HALF tiles (32 keys x 64 rows) across tile boundaries: in step j the matrix pipe computes
//     S(j) and P(j-2)·V while the vector unit exponentiates S(j-1); the instruction order is pinned (sched_barrier) with the
//     vector work spread between the matrix instructions.
//
// All matrix instructions are inline asm (explicit VGPR / AGPR operand classes); hipcc pads no hazard around them, so the
// schedule keeps every MFMA result a whole step away from its first vector reader, and the rare paths pad by hand.
#include "common.h"

#define DH 64
#define TK 64
#define MAX_KTILES 512
#define F64_NST 4
#define F64_STAGE (TK * DH * 2 + TK * 16)          // u16 per stage: K | V | one-hot (18 KiB)
#define F64_THR 12.f                                // log2 units a new score may exceed the running reference maximum by (knob 15 = t + 1: t)

typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

MCA_TRACE_BUFFER(attn_fwd64)      // trace build, knob 8 = 8: s_memtime stamps of wavefront 0 of one workgroup (tools/trace_attn_fwd64.py)
#ifdef MCA_TRACE_BUILD            // stamps are kept in LDS while the loop runs (a global store would count in the ring's vmcnt)
#define F64_STAMP()                                                                                         \
  do {                                                                                                      \
    unsigned long long t_;                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (tracing && ti < 1000) trace_s[ti++] = t_;                                                           \
  } while (0)
#define F64_TRACE_LDS 8192
#define F64_TRACE_PARAMS , unsigned long long* trace_s, int& ti, const bool tracing
#define F64_TRACE_ARGS , trace_s, ti, tracing
#else
#define F64_STAMP() do { } while (0)
#define F64_TRACE_LDS 0
#define F64_TRACE_PARAMS
#define F64_TRACE_ARGS
#endif

#define F64_SB() __builtin_amdgcn_sched_barrier(0)
// register classes: what the vector unit touches (scores, P, O, -m) lives in VGPRs; what only the matrix pipe reads (Q
// fragments for the whole kernel, K / V^T / one-hot fragments straight from LDS, the mask words) lives in the accumulator half
#define MFMA_INIT(D, A, B, C) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(D) : "a"(A), "a"(B), "v"(C))
#define MFMA_ACCV(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "a"(A), "a"(B))
// (the V^T fragments, two 64-bit transposed reads glued into one operand and carried across a step boundary, are kept in
//  VGPRs by hipcc whatever the constraint says: asking for them there saves the copies; operand class does not change the rate)
#define MFMA_ACCA(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "v"(A), "v"(B))
#define MFMA_ACCM(D, A, B) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(D) : "a"(A), "v"(B))

__device__ __forceinline__ int f64_k_off(int r, int c) { return r * 64 + ((c ^ ((r >> 1) & 7)) << 3); }
__device__ __forceinline__ int f64_v_off(int r, int c) { return r * 64 + ((c ^ (((r >> 1) & 1) << 2)) << 3); }

struct f64_lane {          // loop-invariant per-lane LDS offsets (u16 units inside a stage: K at 0, V at 4096, one-hot at 8192)
  int kaddr[4];            // K row fragment of k-step st (key block 0; block 1: + 2048)
  int vaddr[2];            // V^T fragment piece (key 4 lh + tq, d block n); further keys: + 64 per key
  int haddr;               // one-hot row fragment (key block 0; block 1: + 512)
};
// The LDS-DMA is issued by hand (M0 written in the same statement): hipcc then knows of no LDS write in flight and neither
// drains vmcnt(0) in front of the transposed fragment reads (whose intrinsic carries no alias information) nor miscounts; the
// kernel's own counted s_waitcnt vmcnt + s_barrier order the ring.  Fragment reads stay compiler-issued and compiler-counted.
#define F64_DMA16(BASE, OFF, LDS_DST)                                                                             \
  do {                                                                                                            \
    unsigned keep_;                                                                                               \
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep_) : "v"(OFF), "s"(BASE), "s"(LDS_DST) : "memory");                                   \
  } while (0)

struct f64_frags { bf16x8 kf[4]; bf16x8 hf; bf16x8 vf[4]; };          // fragments of one step (accumulator half)
// LDS-DMA of one tile: wave-uniform 64-bit bases + this lane's byte offsets (two K / two V pieces per wavefront, one one-hot
// piece from wavefronts 0 and 1), LDS byte address of this wavefront's first K piece
struct f64_dma { const u16* kb; const u16* vb; const u16* hb; unsigned ko[2], vo[2], ho, lds; bool on, hot; };

// One pipeline step.  DO_S: S(new half) = K.Q^T + mask - m into snew;  DO_SM: exponentiate sx -> pnew (+ row sums, lazy maximum;
// FIRST: the rows' first scores);  DO_PV: O^T += V^T . pold.  `cur` holds the step's fragments (read during the step before);
// RD_K / RD_V: prefetch the NEXT step's K + one-hot / V^T fragments (half HN of the tiles at nkst / nvst) into `nxt`, one read
// per slot;  DMA: issue the pieces of `job` (the tile two ahead), one per third slot.
// The mask product runs on every tile (a tile that needs none adds exact zeros): no branch inside the pinned sequence.
template <bool DO_S, bool DO_SM, bool DO_PV, bool FIRST, int HN, bool RD_K, bool RD_V, bool DMA>
__device__ __forceinline__ void f64_step(f32x16 (&o)[2][2], f32x16 (&negm)[2], float (&l)[2], f32x16 (&snew)[2], f32x16 (&sx)[2],
                                         u32x4v (&pnew)[2][2], const u32x4v (&pold)[2][2], const bf16x8 (&qf)[2][4],
                                         const u32x4v (&qbw)[2], const f64_lane& ln, const f64_frags& cur, f64_frags& nxt,
                                         const u16* nkst, const u16* nvst, const f64_dma& job, const float thr F64_TRACE_PARAMS) {
  // ---- phase 0: block maxima of the pending scores (vector), then the first half of the pending P.V (matrix)
  float mx[2] = {0.f, 0.f};
  if (DO_SM) {
#pragma unroll
    for (int qb = 0; qb < 2; qb++) {
      float m = fmaxf(fmaxf(sx[qb][0], sx[qb][1]), sx[qb][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, sx[qb][r]), sx[qb][r + 1]);
      mx[qb] = fmaxf(m, sx[qb][15]);
    }
  }
  F64_SB();
  if (DO_S && DO_SM && DO_PV && !DMA) F64_STAMP();
  if (DO_PV) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      MFMA_ACCA(o[0][k & 1], cur.vf[k], pold[0][k >> 1]);
      MFMA_ACCA(o[1][k & 1], cur.vf[k], pold[1][k >> 1]);
    }
  }
  F64_SB();
  // ---- decision: does any row's new maximum leave the window?  (wave-uniform; a row block's first scores always do.)
  // Slow path, first half: the pending scores, -m and l move to the new reference maximum NOW (before any exponential and
  // before S(new) takes -m as its start value); O keeps the old one until the step's P.V - exponentiated against the old
  // maximum - has been added (second half, at the end of the step).  Everything at the old scale is rescaled exactly once.
  bool slow = false;
  float alpha[2] = {1.f, 1.f};
  bf16x4 vlo[4];
  if (DO_SM) {
    // a row's reference only ever tracks REAL keys: blocked scores (-32768 from the mask product) underflow to exactly 0
    // against any reference a real key can set, and a reference near -32768 would cost the scores 2^-8 of absolute precision.
    // A row that has accumulated nothing yet (l == 0) takes its reference from the first real score it meets, whatever its sign.
    const bool over = (mx[0] > thr) | (mx[1] > thr) | ((l[0] == 0.f) & (mx[0] > -16384.f)) | ((l[1] == 0.f) & (mx[1] > -16384.f));
    slow = FIRST || __builtin_amdgcn_ballot_w64(over) != 0ull;
    if (__builtin_expect(slow, FIRST)) {          // (the fast path must be the fall-through: a taken branch refills the instruction buffer)
#pragma unroll
      for (int qb = 0; qb < 2; qb++) {
        const float mfull = fmaxf(mx[qb], __shfl_xor(mx[qb], 32, WAVE));
        const float lrow = l[qb] + __shfl_xor(l[qb], 32, WAVE);
        const bool empty = !(lrow > 0.f), real = mfull > -16384.f;
        const float delta = (empty && real) ? mfull : (mfull > thr ? mfull : 0.f);
        alpha[qb] = empty ? 0.f : __builtin_amdgcn_exp2f(-delta);          // (an empty row's O and l are exact zeros)
        l[qb] *= alpha[qb];
#pragma unroll
        for (int r = 0; r < 16; r++) { negm[qb][r] -= delta; sx[qb][r] -= delta; }
      }
      F64_SB();
      asm volatile("s_nop 7" ::: "memory");          // vector writes of -m before the matrix pipe reads it
    }
  }
  F64_SB();
  if (DO_S && DO_SM && DO_PV && !DMA) F64_STAMP();
  // ---- 16 slots: one matrix instruction (while any is left), one fragment read of the next step, now and then one DMA piece,
  // one exponential group (2 exp, 2 add, 1 pack).  Matrix order: S(st 0) x2, PV k=2 x2, S(st 1) x2, PV k=3 x2, S(st 2) x2,
  // S(st 3) x2, mask x2.
#define F64_EXPG(G)                                                                                              \
  if (DO_SM) {                                                                                                   \
    constexpr int qb_ = (G) >> 3, i_ = (G) & 7;                                                                  \
    const float p0_ = __builtin_amdgcn_exp2f(sx[qb_][2 * i_]), p1_ = __builtin_amdgcn_exp2f(sx[qb_][2 * i_ + 1]); \
    l[qb_] += p0_; l[qb_] += p1_;                                                                                \
    pnew[qb_][i_ >> 2][i_ & 3] = pack2bf_pk(p0_, p1_);                                                           \
    /* the group's results are consumed HERE: no pass may sink the group towards their next-step readers */      \
    asm volatile("" :: "v"(pnew[qb_][i_ >> 2][i_ & 3]), "v"(l[qb_]));                                            \
  }                                                                                                              \
  F64_SB();
#define F64_S(ST, QB)                                                                              \
  if (DO_S) { if ((ST) == 0) MFMA_INIT(snew[QB], cur.kf[ST], qf[QB][ST], negm[QB]); else MFMA_ACCV(snew[QB], cur.kf[ST], qf[QB][ST]); }
#define F64_PV(K, QB) if (DO_PV) MFMA_ACCA(o[QB][(K) & 1], cur.vf[K], pold[QB][(K) >> 1]);
#define F64_RK(ST) if (RD_K) nxt.kf[ST] = *reinterpret_cast<const bf16x8*>(nkst + ln.kaddr[ST] + HN * 2048);
#define F64_RH() if (RD_K) nxt.hf = *reinterpret_cast<const bf16x8*>(nkst + 2 * TK * DH + ln.haddr + HN * 512);
  // half T (keys 8 T .. of the 16-key step KS) of V^T fragment K = 2 KS + N: elements 4 T .. 4 T + 3
#define F64_RV(K, T)                                                                                              \
  if (RD_V) {                                                                                                     \
    const bf16x4 v4_ = lds_read_tr16(nvst + TK * DH + ln.vaddr[(K) & 1] + (HN * 32 + 16 * ((K) >> 1) + 8 * (T)) * 64); \
    if ((T) == 0) vlo[K] = v4_; else nxt.vf[K] = __builtin_shufflevector(vlo[K], v4_, 0, 1, 2, 3, 4, 5, 6, 7);     \
  }
#define F64_DK(I) if (DMA) { if (job.on) F64_DMA16(job.kb, job.ko[I], job.lds + (unsigned)(I) * 4096u); }
#define F64_DV(I) if (DMA) { if (job.on) F64_DMA16(job.vb, job.vo[I], job.lds + 8192u + (unsigned)(I) * 4096u); }
#define F64_DH() if (DMA) { if (job.hot) F64_DMA16(job.hb, job.ho, job.lds + 16384u); }
  F64_S(0, 0) F64_RV(0, 0) F64_EXPG(0)
  F64_S(0, 1) F64_RV(0, 1) F64_DK(0) F64_EXPG(1)
  F64_PV(2, 0) F64_RV(1, 0) F64_EXPG(2)
  F64_PV(2, 1) F64_RV(1, 1) F64_EXPG(3)
  F64_S(1, 0) F64_RK(0) F64_DV(0) F64_EXPG(4)
  F64_S(1, 1) F64_RK(1) F64_EXPG(5)
  F64_PV(3, 0) F64_RK(2) F64_EXPG(6)
  F64_PV(3, 1) F64_RK(3) F64_DK(1) F64_EXPG(7)
  if (DO_S && DO_SM && DO_PV && !DMA) F64_STAMP();
  F64_S(2, 0) F64_RH() F64_EXPG(8)
  F64_S(2, 1) F64_RV(2, 0) F64_EXPG(9)
  F64_S(3, 0) F64_RV(2, 1) F64_DV(1) F64_EXPG(10)
  F64_S(3, 1) F64_RV(3, 0) F64_EXPG(11)
  if (DO_S) MFMA_ACCM(snew[0], cur.hf, qbw[0]);
  F64_RV(3, 1) F64_EXPG(12)
  if (DO_S) MFMA_ACCM(snew[1], cur.hf, qbw[1]);
  F64_DH() F64_EXPG(13)
  F64_EXPG(14)
  F64_EXPG(15)
  // the step's own fragments stay live up to here, so that no read of the NEXT step's fragments lands in a register an
  // in-flight matrix instruction still reads (the hardware holds such a read back behind the instruction)
  if (DO_S) asm volatile("" :: "a"(cur.kf[0]), "a"(cur.kf[1]), "a"(cur.kf[2]), "a"(cur.kf[3]), "a"(cur.hf));
  if (DO_PV) asm volatile("" :: "v"(cur.vf[0]), "v"(cur.vf[1]), "v"(cur.vf[2]), "v"(cur.vf[3]));
#undef F64_EXPG
#undef F64_S
#undef F64_PV
#undef F64_RK
#undef F64_RH
#undef F64_RV
#undef F64_DK
#undef F64_DV
#undef F64_DH
  // without exponential groups nothing separates the last S instruction from the next step's first reader of its result
  if (DO_S && !DO_SM) { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); F64_SB(); }
  // ---- slow path, second half: O joins the new reference maximum (the step's P.V is complete: pad for its last result)
  if (DO_SM && !FIRST) {
    if (__builtin_expect(slow, false)) {
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      F64_SB();
#pragma unroll
      for (int qb = 0; qb < 2; qb++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
          for (int r = 0; r < 16; r++) o[qb][n][r] *= alpha[qb];
      F64_SB();
      asm volatile("s_nop 7" ::: "memory");          // accumulator writes before the matrix pipe reads them
      F64_SB();
    }
  }
}

__global__ __launch_bounds__(256, 1) void attn_fwd64_kernel(mca_attn_fwd_args a, int dbg, float thr) {
  // ONE LDS object (a second one beside an LDS-DMA target makes hipcc drain vmcnt before every fragment read)
  extern __shared__ __attribute__((aligned(16))) u16 lds[];          // F64_LDS_BYTES
  uint32_t* live_s = reinterpret_cast<uint32_t*>(lds + F64_NST * F64_STAGE);
  uint8_t* flags_s = reinterpret_cast<uint8_t*>(live_s + MAX_KTILES);
  int* n_live_s = reinterpret_cast<int*>(flags_s + MAX_KTILES);
#ifdef MCA_TRACE_BUILD
  unsigned long long* trace_s = reinterpret_cast<unsigned long long*>(n_live_s + 4);
  const bool tracing = (dbg & 8) && blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z) == 24 && threadIdx.x == 0;
  int ti = 0;
#endif

  const int lin0 = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
  const int lin = (dbg & 16) ? lin0 : xcd_remap(lin0, (int)(gridDim.x * gridDim.y * gridDim.z));
  const int4 desc = reinterpret_cast<const int4*>(a.qb_desc)[lin % (int)gridDim.x];          // slots sorted by descending work
  const int row0 = desc.x, nrows = desc.y, lb = desc.z, le = desc.z + desc.w;
  const int h = (lin / (int)gridDim.x) % (int)gridDim.y, b = lin / (int)(gridDim.x * gridDim.y);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

  int qrow[2];
  bool qvalid[2];
  bf16x8 qf[2][4];
  u32x4v qbw[2];
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    int r = wave * 64 + qb * 32 + l31;
    qvalid[qb] = r < nrows;
    if (r > nrows - 1) r = nrows - 1;
    qrow[qb] = row0 + r;
    const u16* qp = a.q + (int64_t)b * a.q_bstride + (int64_t)qrow[qb] * a.q_ld + h * DH + 8 * lh;
#pragma unroll
    for (int s = 0; s < 4; s++) qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    const uint32_t qm = a.qmask[qrow[qb]];
    const uint32_t qm8 = ((qm >> (8 * lh)) & 0xffu) & (lh ? 0x7fu : 0xffu);
#pragma unroll
    for (int w = 0; w < 4; w++)
      qbw[qb][w] = (((qm8 >> (2 * w)) & 1u) ? 0u : 0xC700u) | (((qm8 >> (2 * w + 1)) & 1u) ? 0u : 0xC7000000u);
  }

  const u16* kbase = a.k + (int64_t)b * a.kv_bstride + h * DH;
  const u16* vbase = a.v + (int64_t)b * a.kv_bstride + h * DH;
  const u16* khot_g = a.khot + (int64_t)b * a.nk_pad * 16;
  {
    const uint8_t* flags_g = a.ktile_flags + (int64_t)b * a.n_ktiles;
    for (int i = tid; i < a.n_ktiles; i += 256) flags_s[i] = flags_g[i];
  }
  __syncthreads();
  // this block's key tiles minus those whose keys are all padded in this sample; bit 31 = no mask product needed
  if (wave == 0) {
    int n = 0;
    for (int i0 = lb; i0 < le; i0 += 64) {
      const int i = i0 + lane;
      const uint32_t e = i < le ? a.qb_kt[i] : 0u;
      const uint8_t fl = i < le ? flags_s[e & 0x7fffffffu] : (uint8_t)0;
      const bool keep = fl != 0;
      const unsigned long long m = __ballot(keep);
      if (keep) live_s[n + __popcll(m & ((1ull << lane) - 1ull))] = (e & 0x7fffffffu) | (((e >> 31) && fl == 2) ? 0x80000000u : 0u);
      n += __popcll(m);
    }
    if (lane == 0) *n_live_s = n;
  }
  __syncthreads();
  const int n_live = __builtin_amdgcn_readfirstlane(*n_live_s);

  // DMA pieces (as attention_fwd.hip's LDS-DMA kernel: the XOR swizzles sit on the SOURCE column, the image is lane-linear)
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
  const unsigned lds_b = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const u16*)lds;
  const unsigned hoff = (unsigned)tid * 16u;          // bytes into a tile's one-hot image (wavefronts 0 and 1)
  auto make_job = [&](int kt, int st, bool on) {
    f64_dma j;
    j.kb = kbase + (int64_t)kt * TK * a.kv_ld;
    j.vb = vbase + (int64_t)kt * TK * a.kv_ld;
    j.hb = khot_g + (int64_t)kt * (TK * 16);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      unsigned ko = koff[i], vo = voff[i];
      if (kt == last_kt) {          // rows past nk: re-read the last valid row (blocked by its one-hot row)
        const int over = kt * TK + prow[i] - (a.nk - 1);
        if (over > 0) { ko -= (unsigned)(over * (int)a.kv_ld); vo -= (unsigned)(over * (int)a.kv_ld); }
      }
      j.ko[i] = ko * 2u; j.vo[i] = vo * 2u;          // bytes
    }
    j.ho = hoff;
    j.lds = lds_b + (unsigned)st * (F64_STAGE * 2u) + (unsigned)wave * 1024u;
    j.on = on; j.hot = on && wave < 2;
    return j;
  };
  auto issue_all = [&](const f64_dma& j) {          // (prologue: the whole tile at once)
    F64_DMA16(j.kb, j.ko[0], j.lds); F64_DMA16(j.vb, j.vo[0], j.lds + 8192u);
    F64_DMA16(j.kb, j.ko[1], j.lds + 4096u); F64_DMA16(j.vb, j.vo[1], j.lds + 8192u + 4096u);
    if (j.hot) F64_DMA16(j.hb, j.ho, j.lds + 16384u);
  };

  f64_lane ln;
  {
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
#pragma unroll
    for (int st = 0; st < 4; st++) ln.kaddr[st] = f64_k_off(l31, 2 * st + lh);
#pragma unroll
    for (int n = 0; n < 2; n++) {
      const int d = n * 32 + 16 * tg + 4 * tp;
      ln.vaddr[n] = f64_v_off(4 * lh + tq, d >> 3) + (d & 7);
    }
    ln.haddr = l31 * 16 + 8 * lh;
  }

  f32x16 o[2][2], negm[2], se[2], so[2];
  u32x4v pe[2][2], po[2][2];
  float l[2] = {0.f, 0.f};
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
#pragma unroll
    for (int r = 0; r < 16; r++) { o[qb][0][r] = 0.f; o[qb][1][r] = 0.f; negm[qb][r] = 0.f; se[qb][r] = 0.f; so[qb][r] = 0.f; }
#pragma unroll
    for (int w = 0; w < 4; w++) { pe[qb][0][w] = 0u; pe[qb][1][w] = 0u; po[qb][0][w] = 0u; po[qb][1][w] = 0u; }
  }

  // ---- the pipeline.  Iteration `it`: even step (S of keys 0-31 of tile it | P.V and exponentials of tile it-1), the tile's
  // ONE sync (tile it+1 has landed for everybody; everybody is past iteration it-1, so the stage of tile it+2 is free), odd
  // step (S of keys 32-63 | ... | prefetch of tile it+1's first fragments | DMA pieces of tile it+2).
  f64_frags fa, fb;
  const f64_dma nojob = make_job(0, 0, false);
  if (n_live > 0) {
    issue_all(make_job((int)(live_s[0] & 0x7fffffffu), 0, true));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (n_live > 1) issue_all(make_job((int)(live_s[1] & 0x7fffffffu), 1, true));
    uint32_t e_next = n_live > 2 ? __builtin_amdgcn_readfirstlane(live_s[2]) : 0u;          // tile whose DMA the next odd step issues
    {          // tile 0's first fragments (the only exposed LDS latency of the block)
#pragma unroll
      for (int st = 0; st < 4; st++) fa.kf[st] = *reinterpret_cast<const bf16x8*>(lds + ln.kaddr[st]);
      fa.hf = *reinterpret_cast<const bf16x8*>(lds + 2 * TK * DH + ln.haddr);
    }
    // tile 0: S_e, then S_o beside the rows' first exponentials (which set the reference maximum)
    f64_step<true, false, false, false, 1, true, false, false>(o, negm, l, se, so, po, pe, qf, qbw, ln, fa, fb, lds, lds, nojob, thr F64_TRACE_ARGS);
    if (n_live > 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    {
      const f64_dma job = make_job((int)(e_next & 0x7fffffffu), 2, n_live > 2);
      e_next = n_live > 3 ? __builtin_amdgcn_readfirstlane(live_s[3]) : 0u;
      f64_step<true, true, false, true, 0, true, true, true>(o, negm, l, so, se, pe, po, qf, qbw, ln, fb, fa, lds + F64_STAGE, lds, job, thr F64_TRACE_ARGS);
    }
    for (int it = 1; it < n_live; it++) {
      F64_STAMP();
      const u16* kst = lds + (it & 3) * F64_STAGE;
      const u16* vst = lds + ((it - 1) & 3) * F64_STAGE;
      // even step: S_e = S(it, keys 0-31) | P_e(it-1, keys 0-31).V | exponentiate S_o(it-1, keys 32-63) -> P_o
      f64_step<true, true, true, false, 1, true, true, false>(o, negm, l, se, so, po, pe, qf, qbw, ln, fa, fb, kst, vst, nojob, thr F64_TRACE_ARGS);
      F64_STAMP();
      if (it + 1 < n_live) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); F64_STAMP(); __builtin_amdgcn_s_barrier(); }
      F64_STAMP();
      const f64_dma job = make_job((int)(e_next & 0x7fffffffu), (it + 2) & 3, it + 2 < n_live);
      e_next = it + 3 < n_live ? __builtin_amdgcn_readfirstlane(live_s[it + 3]) : 0u;
      F64_STAMP();
      // odd step: S_o = S(it, keys 32-63) | P_o(it-1, keys 32-63).V | exponentiate S_e(it, keys 0-31) -> P_e
      f64_step<true, true, true, false, 0, true, true, true>(o, negm, l, so, se, pe, po, qf, qbw, ln, fb, fa, lds + ((it + 1) & 3) * F64_STAGE, kst, job, thr F64_TRACE_ARGS);
      F64_STAMP();
    }
    // drain: the last tile's second half and its two P.V
    const u16* vst = lds + ((n_live - 1) & 3) * F64_STAGE;
    f64_step<false, true, true, false, 1, false, true, false>(o, negm, l, se, so, po, pe, qf, qbw, ln, fa, fb, vst, vst, nojob, thr F64_TRACE_ARGS);
    f64_step<false, false, true, false, 0, false, false, false>(o, negm, l, so, se, pe, po, qf, qbw, ln, fb, fa, vst, vst, nojob, thr F64_TRACE_ARGS);
  }
#ifdef MCA_TRACE_BUILD
  if (tracing) { for (int i = 0; i < ti; i++) mca_trace_attn_fwd64[i] = trace_s[i]; mca_trace_attn_fwd64[1023] = (unsigned long long)ti; }
#endif
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the last MFMA's result before the epilogue reads O
  F64_SB();

  // ---- epilogue (a row that met blocked keys only: no tile at all, or a reference maximum below -16384)
  const float* vm = a.vmean + (int64_t)b * a.heads * DH + h * DH;
#pragma unroll
  for (int qb = 0; qb < 2; qb++) {
    const float lt = l[qb] + __shfl_xor(l[qb], 32, WAVE);
    const float m_run = -negm[qb][0];
    const bool uniform = !(lt > 0.f) || m_run < -16384.f;
    const float inv = uniform ? 0.f : 1.f / lt;
    if (qvalid[qb]) {
      if (lh == 0) a.lse[((int64_t)b * a.heads + h) * a.nq + qrow[qb]] = uniform ? INFINITY : m_run + log2f(lt);
      u16* op = a.o + (int64_t)b * a.o_bstride + (int64_t)qrow[qb] * a.o_ld + h * DH;
#pragma unroll
      for (int n = 0; n < 2; n++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const int d = n * 32 + 8 * g + 4 * lh;
          float v0, v1, v2, v3;
          if (uniform) { v0 = vm[d]; v1 = vm[d + 1]; v2 = vm[d + 2]; v3 = vm[d + 3]; }
          else { v0 = o[qb][n][4 * g] * inv; v1 = o[qb][n][4 * g + 1] * inv; v2 = o[qb][n][4 * g + 2] * inv; v3 = o[qb][n][4 * g + 3] * inv; }
          uint2 pk; pk.x = pack2bf(v0, v1); pk.y = pack2bf(v2, v3);
          *reinterpret_cast<uint2*>(op + d) = pk;
        }
    }
  }
}

// launched by mca_attn_fwd (attention_fwd.hip) when the caller supplies the query-block schedule
#define F64_LDS_BYTES ((F64_NST * F64_STAGE + 2 * MAX_KTILES + MAX_KTILES / 2 + 8) * 2 + F64_TRACE_LDS)
int mca_attn_fwd64_launch(const mca_attn_fwd_args* a, hipStream_t stream, int dbg) {
  if (!a->qb_desc || !a->qb_kt || a->n_qblocks <= 0 || !a->khot) return MCA_E_BADARG;
  if ((uintptr_t)a->qb_desc % 16) return MCA_E_ALIGN;
  static bool attr_set[64] = {false};          // hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute
  bool* done = mca_dev_flag(attr_set);
  if (!*done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F64_LDS_BYTES) != hipSuccess)
      return MCA_E_LAUNCH;
    *done = true;
  }
  const dim3 grid(a->n_qblocks, a->heads, a->batch);
  const float thr = mca_knobs[15] > 0 ? (float)(mca_knobs[15] - 1) : F64_THR;          // (tests: the rescale branch at every tile)
  hipLaunchKernelGGL(attn_fwd64_kernel, grid, dim3(256), F64_LDS_BYTES, stream, *a, dbg, thr);
  return launch_status();
}

"""Static sparsity structure of the MCA / MMA (Zorro) fusion transformer.

The reference materialises two dense boolean masks at init time (``attn_mask`` (N,N) and ``pool_mask``
(R,N), /root/reference/model.py:355-372,383-446) and broadcasts them to (b,h,N,N) on every call.  Here
the same structure is kept in a compact *group* form the HIP kernels consume directly:

  * every key token j has a group id ``kgroup[j]`` (modality m -> m; fusion sub-block c -> M+c, or the
    single group M under Zorro),
  * every query row i has a 32-bit set ``qmask[i]`` of the key groups it may attend,
  * ``allowed(i, j) = (qmask[i] >> kgroup[j]) & 1``; bit 31 is reserved: a padded key gets id 31 at run
    time and no query ever has bit 31 set.

From that the host derives the tile schedule (which 64-key tiles a 128-row query tile has to visit and
whether a tile needs element-wise masking at all), so the attention kernels skip the ~57 % of the score
matrix the masks rule out instead of computing and discarding it.
The dense masks are still produced (``dense_attn_mask`` / ``dense_pool_mask``) because they are
persistent buffers in the reference's state_dict.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from itertools import combinations
from typing import Dict, List, Sequence

import numpy as np

FUSION_TOKEN = -1
GLOBAL_TOKEN = -2
PAD_GROUP = 31          # run-time group id of a padded key; never present in any qmask
MAX_GROUPS = 31


def combos_of(n_modalities: int, powers: Sequence[int]) -> List[frozenset]:
    """Ordered fusion-channel combinations: for each cardinality in ``powers`` (in the given order) all
    modality subsets of that size in lexicographic order (reference: model.py:11-12,312)."""
    out: List[frozenset] = []
    for r in powers:
        out.extend(frozenset(c) for c in combinations(range(n_modalities), r))
    return out


@dataclass
class TileSchedule:
    """CSR list of key tiles per query tile (forward) and query tiles per key tile (backward)."""
    bq: int
    bk: int
    n_q: int
    n_k: int
    q_ptr: np.ndarray       # (nQ+1,) int32
    q_kt: np.ndarray        # (nnz,) int32   key-tile index
    q_full: np.ndarray      # (nnz,) uint8   1 = every (i,j) of the tile is structurally allowed
    q_order: np.ndarray     # (nQ,) int32    query tiles sorted by descending work (launch order)
    k_ptr: np.ndarray       # (nK+1,) int32
    k_qt: np.ndarray        # (nnz,) int32
    k_full: np.ndarray      # (nnz,) uint8
    k_order: np.ndarray     # (nK,) int32
    allowed_pairs: int      # number of structurally allowed (i,j) pairs  (algorithmic work)
    visited_pairs: int      # pairs in visited tiles (executed work, including partial-tile waste)


def build_schedule(qmask: np.ndarray, kgroup: np.ndarray, bq: int, bk: int) -> TileSchedule:
    nq_rows, nk_rows = len(qmask), len(kgroup)
    allowed = ((qmask[:, None].astype(np.uint32) >> kgroup[None, :].astype(np.uint32)) & 1).astype(bool)
    nQ, nK = -(-nq_rows // bq), -(-nk_rows // bk)
    any_t = np.zeros((nQ, nK), bool)
    all_t = np.zeros((nQ, nK), bool)
    for qi in range(nQ):
        rows = allowed[qi * bq:(qi + 1) * bq]
        for ki in range(nK):
            blk = rows[:, ki * bk:(ki + 1) * bk]
            any_t[qi, ki] = blk.any()
            # a tile that runs past the end of the keys always needs the element-wise path
            all_t[qi, ki] = blk.all() and (ki + 1) * bk <= nk_rows
    q_ptr = np.zeros(nQ + 1, np.int32)
    q_kt, q_full = [], []
    for qi in range(nQ):
        ks = np.nonzero(any_t[qi])[0]
        q_kt.extend(ks.tolist())
        q_full.extend(all_t[qi, ks].astype(np.uint8).tolist())
        q_ptr[qi + 1] = len(q_kt)
    k_ptr = np.zeros(nK + 1, np.int32)
    k_qt, k_full = [], []
    for ki in range(nK):
        qs = np.nonzero(any_t[:, ki])[0]
        k_qt.extend(qs.tolist())
        k_full.extend(all_t[qs, ki].astype(np.uint8).tolist())
        k_ptr[ki + 1] = len(k_qt)
    q_work = np.diff(q_ptr)
    k_work = np.diff(k_ptr)
    return TileSchedule(
        bq=bq, bk=bk, n_q=nQ, n_k=nK,
        q_ptr=q_ptr, q_kt=np.asarray(q_kt, np.int32), q_full=np.asarray(q_full, np.uint8),
        q_order=np.argsort(-q_work, kind="stable").astype(np.int32),
        k_ptr=k_ptr, k_qt=np.asarray(k_qt, np.int32), k_full=np.asarray(k_full, np.uint8),
        k_order=np.argsort(-k_work, kind="stable").astype(np.int32),
        allowed_pairs=int(allowed.sum()),
        visited_pairs=int(any_t.sum()) * bq * bk,
    )


@dataclass
class BlockSchedule:
    """Query blocks of up to ``rows`` rows cut ALONG the structure (mca_attn_fwd_args.qb_desc / qb_kt): one workgroup each."""
    rows: int
    bk: int
    desc: np.ndarray        # (nB, 4) int32 {first row, rows, first list entry, entries}, launch order (descending work)
    kt: np.ndarray          # (nnz,) uint32 key-tile index | (structurally full << 31)
    allowed_pairs: int
    visited_pairs: int      # rows * bk per list entry (executed work of a full-width workgroup)


def build_block_schedule(qmask: np.ndarray, kgroup: np.ndarray, rows: int = 256, bk: int = 64) -> BlockSchedule:
    """Cut the query rows into blocks that follow the structure: a run of rows with one query mask (a modality, the Zorro
    fusion block) of at least rows / 2 rows is split evenly into ceil(len / rows) blocks of its own; shorter runs (a small
    modality, the MCA fusion sub-blocks) are merged with their neighbours up to ``rows`` rows.  A block of a large modality
    then visits that modality's key tiles only, and almost all of them need no mask."""
    n, nk = len(qmask), len(kgroup)
    runs, start = [], 0
    for i in range(1, n + 1):
        if i == n or qmask[i] != qmask[start]:
            runs.append((start, i - start))
            start = i
    blocks, pend = [], None
    for r0, ln in runs:
        if ln >= rows // 2:
            if pend:
                blocks.append(tuple(pend)); pend = None
            nch = -(-ln // rows)
            base, rem, at = ln // nch, ln % nch, r0
            for c in range(nch):
                sz = base + (1 if c < rem else 0)
                blocks.append((at, sz)); at += sz
        else:
            if pend and pend[1] + ln <= rows and pend[0] + pend[1] == r0:
                pend[1] += ln
            else:
                if pend:
                    blocks.append(tuple(pend))
                pend = [r0, ln]
    if pend:
        blocks.append(tuple(pend))
    allowed = ((qmask[:, None].astype(np.uint32) >> kgroup[None, :].astype(np.uint32)) & 1).astype(bool)
    nK = -(-nk // bk)
    lists = []
    for r0, ln in blocks:
        sub = allowed[r0:r0 + ln]
        ent = []
        for ki in range(nK):
            blk = sub[:, ki * bk:(ki + 1) * bk]
            if blk.any():
                full = bool(blk.all()) and (ki + 1) * bk <= nk
                ent.append(np.uint32(ki) | (np.uint32(1 << 31) if full else np.uint32(0)))
        lists.append(ent)
    order = sorted(range(len(blocks)), key=lambda i: -len(lists[i]))
    desc, kt = [], []
    for i in order:
        desc.append((blocks[i][0], blocks[i][1], len(kt), len(lists[i])))
        kt.extend(lists[i])
    assert sum(b[1] for b in blocks) == n
    return BlockSchedule(rows=rows, bk=bk, desc=np.asarray(desc, np.int32).reshape(-1, 4), kt=np.asarray(kt, np.uint32),
                         allowed_pairs=int(allowed.sum()), visited_pairs=len(kt) * rows * bk)


@dataclass
class OnePassSchedule:
    """Work list of the one-pass attention backward (mca_attn_bwd_onepass, attention_bwd1.hip): ONE workgroup owns a whole
    (sample, head), walks the key blocks in order and, inside a key block, the query tiles the structure allows."""
    tq: int
    tk: int
    qt_desc: np.ndarray     # (nQ, 2) int32 {first row, rows <= tq}
    kb_desc: np.ndarray     # (nK, 4) int32 {first key, keys <= tk, first list entry, entries}
    kb_qt: np.ndarray       # (nnz,) uint32 query tile | (every pair of the tile structurally allowed << 31)
    visit: np.ndarray       # (nK, nQ) uint8: the key block lists the query tile
    row_slot: np.ndarray    # (n,) int32: tile * tq + position of the row inside its tile
    allowed_pairs: int
    visited_pairs: int      # tq * tk per list entry (executed work of a full-width step)


def _cut_runs(runs, cap: int):
    """Cut [(start, length)] runs of equal structure into pieces of at most ``cap`` rows: a run of at least cap / 2 rows is
    split evenly into pieces of its own, shorter neighbours are merged up to ``cap``."""
    pieces, pend = [], None
    for r0, ln in runs:
        if ln >= cap // 2:
            if pend:
                pieces.append(tuple(pend)); pend = None
            nch = -(-ln // cap)
            base, rem, at = ln // nch, ln % nch, r0
            for c in range(nch):
                sz = base + (1 if c < rem else 0)
                pieces.append((at, sz)); at += sz
        elif pend and pend[1] + ln <= cap and pend[0] + pend[1] == r0:
            pend[1] += ln
        else:
            if pend:
                pieces.append(tuple(pend))
            pend = [r0, ln]
    if pend:
        pieces.append(tuple(pend))
    return pieces


def build_onepass_schedule(qmask: np.ndarray, kgroup: np.ndarray, tq: int = 64, tk: int = 256, aligned: bool = True,
                           serpentine: bool = True) -> OnePassSchedule:
    """Query tiles of at most ``tq`` rows and key blocks of at most ``tk`` keys.  aligned: both are cut ALONG the structure (a
    modality's 1,500 tokens become 24 query tiles of 62-63 rows and 6 key blocks of 250 keys: no tile straddles two modalities,
    CMU: 193 steps of 64 x 256 per (sample, head) against 210 on the plain grid); else the plain grid.
    serpentine: every other key block walks its query tiles in descending order, so the tiles a sweep ends on are the ones the next
    sweep starts with: their Q / dO rows and dQ partials are re-used at a short distance (L2 / Infinity Cache instead of HBM;
    the order of a block's list is free: a tile's first / last visit is decided per key block)."""
    n = len(qmask)
    assert len(kgroup) == n, "self-attention only"
    if aligned:
        runs, start = [], 0
        for i in range(1, n + 1):
            if i == n or qmask[i] != qmask[start] or kgroup[i] != kgroup[start]:
                runs.append((start, i - start)); start = i
        qts, kbs = _cut_runs(runs, tq), _cut_runs(runs, tk)
    else:
        qts = [(r, min(tq, n - r)) for r in range(0, n, tq)]
        kbs = [(r, min(tk, n - r)) for r in range(0, n, tk)]
    assert sum(x[1] for x in qts) == n and sum(x[1] for x in kbs) == n
    allowed = ((qmask[:, None].astype(np.uint32) >> kgroup[None, :].astype(np.uint32)) & 1).astype(bool)
    visit = np.zeros((len(kbs), len(qts)), np.uint8)
    kb_desc, kb_qt = [], []
    for ki, (k0, kn) in enumerate(kbs):
        first = len(kb_qt)
        ent = []
        for qi, (q0, qn) in enumerate(qts):
            blk = allowed[q0:q0 + qn, k0:k0 + kn]
            if blk.any():
                visit[ki, qi] = 1
                ent.append(np.uint32(qi) | (np.uint32(1 << 31) if blk.all() else np.uint32(0)))
        kb_qt.extend(ent[::-1] if (serpentine and ki % 2 == 1) else ent)
        kb_desc.append((k0, kn, first, len(kb_qt) - first))
    row_slot = np.zeros(n, np.int32)
    for qi, (q0, qn) in enumerate(qts):
        row_slot[q0:q0 + qn] = qi * tq + np.arange(qn)
    return OnePassSchedule(tq=tq, tk=tk, qt_desc=np.asarray(qts, np.int32).reshape(-1, 2), kb_desc=np.asarray(kb_desc, np.int32).reshape(-1, 4),
                           kb_qt=np.asarray(kb_qt, np.uint32), visit=visit, row_slot=row_slot,
                           allowed_pairs=int(allowed.sum()), visited_pairs=len(kb_qt) * tq * tk)


@dataclass
class FusionStructure:
    """Everything the reference's ``MCA.__init__`` derives from the config (model.py:305-372)."""
    token_dims: List[int]
    num_fusion_tokens: int
    fusion_combos_powers: Sequence[int] = (4, 5)
    fcl: bool = False
    zorro: bool = False
    no_fusion: bool = False

    combos: List[frozenset] = field(init=False)
    return_token_types: List[int] = field(init=False)
    token_types: np.ndarray = field(init=False)
    kgroup: np.ndarray = field(init=False)
    qmask_attn: np.ndarray = field(init=False)
    qmask_pool: np.ndarray = field(init=False)

    def __post_init__(self):
        M = len(self.token_dims)
        self.combos = combos_of(M, self.fusion_combos_powers)
        C = len(self.combos)
        if self.no_fusion:
            self.num_fusion_tokens = 0
        F = self.num_fusion_tokens
        # return tokens: one per modality, then fusion token(s), then the global token
        if self.no_fusion:
            self.return_token_types = list(range(M)) + [GLOBAL_TOKEN]
        elif (not self.fcl) or self.zorro:
            self.return_token_types = list(range(M)) + [FUSION_TOKEN, GLOBAL_TOKEN]
        else:
            self.return_token_types = list(range(M)) + [FUSION_TOKEN] * C + [GLOBAL_TOKEN]

        types = []
        for m, n in enumerate(self.token_dims):
            types += [m] * n
        types += [FUSION_TOKEN] * F
        self.token_types = np.asarray(types, np.int64)
        N = len(types)

        mca = not self.zorro            # modal-incomplete fusion channels in the attention mask
        if mca and F:
            if F % C != 0:
                raise AssertionError(
                    f"Number of fusion tokens {F} must be divisible by the number of combinations {C}")
            nsub = F // C
            n_groups = M + C
        else:
            nsub = F
            n_groups = M + (1 if F else 0)
        if n_groups > MAX_GROUPS:
            raise NotImplementedError(
                f"{n_groups} key groups exceed the {MAX_GROUPS} the 32-bit query masks of the HIP kernels hold")

        kgroup = np.zeros(N, np.uint8)
        qmask = np.zeros(N, np.uint32)
        off = 0
        for m, n in enumerate(self.token_dims):
            kgroup[off:off + n] = m
            qmask[off:off + n] = 1 << m                      # a modality token sees its own modality only
            off += n
        all_fusion_bits = 0
        if F:
            if mca:
                for c, combo in enumerate(self.combos):
                    g = M + c
                    kgroup[off + c * nsub: off + (c + 1) * nsub] = g
                    bits = (1 << g)
                    for m in combo:
                        bits |= 1 << m
                    qmask[off + c * nsub: off + (c + 1) * nsub] = bits
                    all_fusion_bits |= 1 << g
            else:
                kgroup[off:] = M
                qmask[off:] = (1 << (M + 1)) - 1            # Zorro fusion tokens see everything
                all_fusion_bits = 1 << M
        self.kgroup, self.qmask_attn = kgroup, qmask
        self.n_groups = n_groups
        self.nsub = nsub

        all_bits = (1 << n_groups) - 1
        qp = []
        fusion_seen = 0
        for t in self.return_token_types:
            if t >= 0:
                qp.append(1 << t)
            elif t == GLOBAL_TOKEN:
                qp.append(all_bits)
            else:
                if mca and self.fcl and F:                  # fusion return token c pools sub-block c only
                    qp.append(1 << (M + fusion_seen))
                    fusion_seen += 1
                else:
                    qp.append(all_fusion_bits)
        self.qmask_pool = np.asarray(qp, np.uint32)

    # ---- sizes ---------------------------------------------------------------------------------
    @property
    def n_tokens(self) -> int:
        return len(self.token_types)

    @property
    def n_return(self) -> int:
        return len(self.return_token_types)

    @property
    def n_modalities(self) -> int:
        return len(self.token_dims)

    # ---- dense forms (state_dict buffers; True = blocked) --------------------------------------
    def dense_attn_mask(self) -> np.ndarray:
        return ~(((self.qmask_attn[:, None] >> self.kgroup[None, :].astype(np.uint32)) & 1).astype(bool))

    def dense_pool_mask(self) -> np.ndarray:
        return ~(((self.qmask_pool[:, None] >> self.kgroup[None, :].astype(np.uint32)) & 1).astype(bool))

    # ---- tile schedules ------------------------------------------------------------------------
    def attn_schedule(self, bq: int = 128, bk: int = 64) -> TileSchedule:
        return build_schedule(self.qmask_attn, self.kgroup, bq, bk)

    def pool_schedule(self, bq: int = 32, bk: int = 64) -> TileSchedule:
        return build_schedule(self.qmask_pool, self.kgroup, bq, bk)

    def attn_block_schedule(self, rows: int = 256, bk: int = 64) -> BlockSchedule:
        return build_block_schedule(self.qmask_attn, self.kgroup, rows, bk)

    def attn_onepass_schedule(self, aligned: bool = True) -> OnePassSchedule:
        return build_onepass_schedule(self.qmask_attn, self.kgroup, 64, 256, aligned)


# --------------------------------------------------------------------------------------------------
# EAO baseline: one SEGMENT per modality and per combination                    (model.py:481-596)
# --------------------------------------------------------------------------------------------------
class EAOStructure:
    """The reference's ``EAO.forward`` (model.py:573-596) runs the SAME layer stack once per modality and once per
    combination of modalities, each time on the concatenation of those modalities' tokens with the padding mask as the only
    attention mask, and mean-pools the un-padded tokens of the pass.  Here all passes are segments of one super-sequence:

      * segment s < M holds modality s alone, segment M + c the modalities of ``combos[c]`` (ascending);
      * ``kgroup[j]`` = segment of token j, ``qmask[i]`` = 1 << segment(i): attention is block-diagonal over segments, which
        is what separate passes compute (LayerNorm, the linear layers and GEGLU act row by row);
      * ``copies``: (source offset, destination offset, rows) of every replica of a modality block; the encoders write
        segment m (= the first M segments, offsets as in the fusion model), the replicas are copies and their gradients
        are summed back.
    Duck-types the fields of ``FusionStructure`` the engine and ``loss_terms`` read."""

    def __init__(self, token_dims: Sequence[int], fusion_combos_powers: Sequence[int], fcl: bool, zorro: bool):
        self.token_dims = list(token_dims)
        M = len(self.token_dims)
        self.combos = combos_of(M, fusion_combos_powers)
        self.fcl, self.zorro, self.no_fusion = bool(fcl), bool(zorro), True
        self.num_fusion_tokens = 0
        self.segments: List[List[int]] = [[m] for m in range(M)] + [sorted(c) for c in self.combos]
        if len(self.segments) > MAX_GROUPS:
            raise NotImplementedError(f"{len(self.segments)} EAO passes exceed the {MAX_GROUPS} key groups of the HIP kernels")
        single_off = np.concatenate([[0], np.cumsum(self.token_dims)]).astype(int)
        seg_start, kgroup, types, copies = [0], [], [], []
        for s, mods in enumerate(self.segments):
            for m in mods:
                n = self.token_dims[m]
                if s >= M:
                    copies.append((int(single_off[m]), len(kgroup), n))
                kgroup += [s] * n
                types += [m] * n
            seg_start.append(len(kgroup))
        self.seg_start = np.asarray(seg_start, np.int32)
        self.copies = copies
        self.kgroup = np.asarray(kgroup, np.uint8)
        self.qmask_attn = (np.uint32(1) << self.kgroup.astype(np.uint32)).astype(np.uint32)
        self.qmask_pool = np.zeros(len(self.segments), np.uint32)          # no attentive pooling
        self.token_types_expanded = np.asarray(types, np.int64)
        # the reference's own (unexpanded) token_types buffer: state_dict key `token_types` (model.py:528,540-547)
        self.token_types = np.asarray([m for m, n in enumerate(self.token_dims) for _ in range(n)], np.int64)
        self.return_token_types = list(range(M))

    @property
    def n_tokens(self) -> int:
        return len(self.kgroup)

    @property
    def n_return(self) -> int:          # pooled slots: one per segment
        return len(self.segments)

    @property
    def n_modalities(self) -> int:
        return len(self.token_dims)

    def attn_schedule(self, bq: int = 128, bk: int = 64) -> TileSchedule:
        return build_schedule(self.qmask_attn, self.kgroup, bq, bk)

    def attn_block_schedule(self, rows: int = 256, bk: int = 64) -> BlockSchedule:
        return build_block_schedule(self.qmask_attn, self.kgroup, rows, bk)

    def attn_onepass_schedule(self, aligned: bool = True) -> OnePassSchedule:
        return build_onepass_schedule(self.qmask_attn, self.kgroup, 64, 256, aligned)

    def dense_attn_mask(self) -> np.ndarray:
        """True = blocked (the block-diagonal complement), as FusionStructure.dense_attn_mask"""
        return ~(((self.qmask_attn[:, None] >> self.kgroup[None, :].astype(np.uint32)) & 1).astype(bool))

    @property
    def block_dims(self) -> List[int]:
        """lengths of the modality blocks of the super-sequence, in order (segment by segment)"""
        return [self.token_dims[m] for mods in self.segments for m in mods]


# --------------------------------------------------------------------------------------------------
# loss schedule: which pooled slots are contrasted and which samples count   (model.py:132-233)
# --------------------------------------------------------------------------------------------------
@dataclass
class LossTerm:
    name: str
    slot_a: int
    slot_b: int
    and_bits: int        # every modality in this set must be present in the sample
    or_bits: int         # and at least one of these (0 = no such condition)


def loss_terms(modalities: Sequence[str], st: FusionStructure, bimodal_contrastive: bool,
               non_fusion_fcl: bool) -> List[LossTerm]:
    """Ordered contrastive terms of ``MCAPretrainingLoss`` and their row-mask rules.

    Pair terms: (modality, fusion) masks on the modality; (modality, modality) on both.  Fusion-channel
    terms (only with fcl and not zorro): (fusion, combo) masks on ANY modality of the combo being present;
    (modality, combo) additionally requires that modality.  Names follow model.py:209,215,220."""
    M = len(modalities)
    do_fcl = st.fcl and not st.zorro
    slot: Dict[object, int] = {m: i for i, m in enumerate(modalities)}
    if do_fcl:
        for c, combo in enumerate(st.combos):
            slot[combo] = M + c
        if not st.no_fusion:
            slot["fusion"] = slot[st.combos[0]]
    elif not st.no_fusion:
        slot["fusion"] = M
    if st.no_fusion:
        pairs = list(combinations(modalities, 2))
    elif bimodal_contrastive:
        pairs = list(combinations(list(modalities) + ["fusion"], 2))
    else:
        pairs = [(m, "fusion") for m in modalities]
    bit = {m: 1 << i for i, m in enumerate(modalities)}
    terms: List[LossTerm] = []
    for a, b in pairs:
        need = 0
        for x in (a, b):
            if x != "fusion":
                need |= bit[x]
        terms.append(LossTerm("_".join(sorted((a, b))), slot[a], slot[b], need, 0))
    if do_fcl:
        root = st.combos[0]
        for combo in st.combos:
            if combo == root:
                continue
            cname = "_".join(sorted(modalities[i] for i in combo))
            anyb = 0
            for i in combo:
                anyb |= 1 << i
            if not st.no_fusion:
                terms.append(LossTerm(f"fcl_fusion|{cname}", slot["fusion"], slot[combo], 0, anyb))
            if non_fusion_fcl:
                for m in modalities:
                    terms.append(LossTerm(f"fcl_{m}|{cname}", slot[m], slot[combo], bit[m], anyb))
    return terms

"""Generates mca-paper_amd/csrc/attention_bwd1_sched.inc: the issue order of ONE loop iteration of the one-pass attention
backward (attention_bwd1.hip, attn_bwd1p_kernel) - every matrix instruction, LDS read / write, vector operation and memory
operation of a wavefront assigned to a slot (one slot = one MFMA and the single-issue work that runs in its shadow), as the CDNA
guide's attention-backward notes ask for ("assigned to an MFMA gap by a generated table").

A wavefront owns 64 keys (two 32-key blocks kb) and a step brings 64 query rows (two 32-row blocks qb): four 32 x 32 score
BLOCKS j = 2 qb + kb per step.  Per block:
    A(j)  S^T, dP^T: the mask product, then 4 + 4 MFMAs on row fragments of Q / dO (row constants = accumulator start values)
    V(j)  8 VE2 (exp2 + multiply of two elements) and 8 VC (two packed conversions -> one dword of P, one of dS)
    C(j)  dV^T, dK^T: 8 MFMAs on transposed fragments of dO / Q, B operands = the packed P / dS
    W(j)  dS^T to the [key][query] image (4 stores of 8 bytes)
dQ' = the PREVIOUS step's dQ^T block (16 MFMAs over the workgroup's 256 keys: needs every wavefront's dS^T, i.e. the barrier).

The loop is ROTATED: iteration X runs A0..A3, V0..V2, C0, C1, W0..W2 of step X, and V3', W3', C2', C3' and dQ' of step X-1.
The one barrier of an iteration sits behind W3' (early in the iteration, in the shadow of A0 / C2', whose MFMAs do not depend on
it); behind it come dQ' and the DMA of step X+2 (three stages); the row constants / fragments of step X+1's first block are read
at the end of the iteration (their stage landed before this iteration's barrier).  The dQ block of step X-2 is stored, and the old
partial of step X's tile loaded into the same registers, at the head of the iteration, one instruction every other slot.
    MFMA stream:   A0 | C2' | C3' | A1 | A2 | C0 | A3 | C1      with the 16 dQ' MFMAs spread behind the barrier
The row constants of a block are read STRAIGHT INTO its score accumulators (the start values of the S^T / dP^T chains), so a
block's constants can only be requested once the vector work of the block that used the set before is done (block 1: V3').
Score sets and packed sets alternate (block j uses set j & 1).  Every other item has an EARLIEST slot (its producers: an MFMA
result needs LAG slots before a vector instruction may read it - an inline-asm MFMA gets no hazard padding from hipcc, the distance
IS the padding; a register set is free only behind its last reader) and a DEADLINE (its consumer minus the LDS latency); a list
scheduler places them earliest-deadline-first under an issue-cost budget per slot and fails loudly if a deadline cannot be met.

Usage: python tools/gen_bwd1_schedule.py   (writes the .inc; the kernel defines the macros)."""
import os

LAG = 2          # slots between an MFMA and the first vector instruction that reads its result (>= 11 wait states for 8 passes)
LDS_LAT = 6      # slots between an LDS read and the MFMA that consumes it (measured: with 3 the MFMAs waited 170 us per layer for fragments)
VW = 2           # slots between a vector write of an MFMA operand (packed P / dS) and that MFMA (hipcc pads nothing around inline asm: 2 wait states)
BUDGET = 24      # issue cycles of fillers per slot (an MFMA holds the issue port for 8 of its 32 cycles)
COST = {"VE2": 24, "VC": 8, "VEP2": 24, "VCP": 8, "RC_L": 4, "RC_D": 4, "RQB": 4, "RF_Q": 4, "RF_O": 4, "TR_O": 8, "TR_Q": 8, "DSW": 6,
        "DSWP": 6, "DQR": 16, "BARRIER": 24, "DMA": 12, "ST": 8, "LD": 6, "NRC_L": 4, "NRC_D": 4, "NRQB": 4, "NRF_Q": 4, "NRF_O": 4}


def a_mfmas(j):
    out = [f"A_M({j})"]
    for ks in range(4):
        out += [f"A_S({j}, {ks})", f"A_P({j}, {ks})"]
    return out


def c_mfmas(j, prev=False):
    if prev:
        return [f"CP_{w}({j}, {sp}, {n})" for sp in range(2) for n in range(2) for w in "VK"]
    return [f"C_{w}({j}, {sp}, {n})" for sp in range(2) for n in range(2) for w in "VK"]


def cname(j, w, sp, n):
    return f"CP_{w}({j}, {sp}, {n})" if j >= 2 else f"C_{w}({j}, {sp}, {n})"


DQ_END_GAP = 6         # MFMA slots behind the last dQ' MFMA: its result is stored right behind the stream (12 wait states, without nops)


def spread_from(main, extra, start):
    """the items of `extra` spread evenly through main[start:len(main) - DQ_END_GAP] (the rest of main follows)"""
    keep = main[len(main) - DQ_END_GAP:]
    main = main[:len(main) - DQ_END_GAP]
    out = _spread_from(main, extra, start)
    return out + keep


def _spread_from(main, extra, start):
    head, tail = main[:start], main[start:]
    out, n, m, e = [], len(tail), len(extra), 0
    for i, x in enumerate(tail):
        out.append(x)
        while e < m and (e + 1) * n <= (i + 1) * m:
            out.append(extra[e]); e += 1
    return head + out + extra[e:]


DQ_START = 20          # first stream position a dQ' MFMA may take (behind the barrier + the LDS latency of its fragments)


def mfma_stream():
    dq = [f"DQM({k})" for k in range(16)]
    base = a_mfmas(0) + c_mfmas(2, True) + c_mfmas(3, True) + a_mfmas(1) + a_mfmas(2) + c_mfmas(0) + a_mfmas(3) + c_mfmas(1)
    return spread_from(base, dq, DQ_START)


def build():
    stream = mfma_stream()
    pos = {m: i for i, m in enumerate(stream)}
    n = len(stream)
    end = n          # the pseudo slot behind the last MFMA
    items = {}       # name -> [earliest, deadline]

    def add(name, earliest, deadline):
        assert name not in items, name
        items[name] = [max(0, earliest), deadline]

    # ---- block 3 of the previous step: its scores (set 1) were finished by the previous iteration's A3 (at least LAG slots before
    # its end) and are overwritten by this iteration's A1
    a3_tail = n - 1 - pos["A_P(3, 3)"]
    v3_ready = max(0, LAG - a3_tail)
    for i in range(8):          # (two elements per item: exp, exp, multiply, multiply - a dependent multiply right behind its exp costs a wait state)
        add(f"VEP2({i})", v3_ready, pos["A_M(1)"] - LDS_LAT - 1)
    for i in range(8):
        sp = i >> 2
        # packed set 1: C1, its last reader, ended the previous iteration; next reader C3'
        add(f"VCP({i})", v3_ready, min(pos["A_M(1)"] - LDS_LAT - 1, pos[f"CP_V(3, {sp}, 0)"] - VW, pos[f"CP_K(3, {sp}, 0)"] - VW))
    for sp in range(2):
        for t in range(2):
            add(f"DSWP({sp}, {t})", v3_ready, end)
    add("BARRIER()", 0, DQ_START - LDS_LAT - 1)
    # ---- vector memory, ONE instruction per wavefront every other slot (the four wavefronts of a CU share one address / data path:
    # 16 cycles per 1-KiB instruction; issued in bursts of 8 - 13 the wavefronts queued behind one another with their MFMA pipes
    # idle, ~100 us per layer): the dQ block of step X-2 goes out (ST), the old partial of step X's tile comes in (LD, the same
    # registers, behind the stores), the five DMA pieces of step X+2's stage follow the barrier
    bar_dl = DQ_START - LDS_LAT - 1
    pinned = {}
    for g in range(4):
        pinned[f"ST({g})"] = 2 * g
        pinned[f"LD({g})"] = 8 + 2 * g
    for q in range(5):
        pinned[f"DMA({q})"] = bar_dl + 3 + 2 * q
    # ---- mask operand / row fragments: ONE register set each.  qb 0 of THIS step was read by the previous iteration (N* items);
    # qb 1 follows behind the readers of qb 0 in A1, the next step's qb 0 behind the readers of qb 1 in A3
    add("RQB(1)", pos["A_M(1)"] + 1, pos["A_M(2)"] - LDS_LAT)
    for ks in range(4):
        add(f"RF_Q(1, {ks})", pos[f"A_S(1, {ks})"] + 1, pos[f"A_S(2, {ks})"] - LDS_LAT)
        add(f"RF_O(1, {ks})", pos[f"A_P(1, {ks})"] + 1, pos[f"A_P(2, {ks})"] - LDS_LAT)
    add("NRQB()", pos["A_M(3)"] + 1, n + pos["A_M(0)"] - LDS_LAT)
    for ks in range(4):
        add(f"NRF_Q({ks})", pos[f"A_S(3, {ks})"] + 1, n + pos[f"A_S(0, {ks})"] - LDS_LAT)
        add(f"NRF_O({ks})", pos[f"A_P(3, {ks})"] + 1, n + pos[f"A_P(0, {ks})"] - LDS_LAT)
    # ---- row constants of block j: into the score set j & 1, behind the vector work of the block that held it (dependencies below)
    for j in (1, 2, 3):
        for g in range(4):
            add(f"RC_L({j}, {g})", 0, pos[f"A_M({j})"] - LDS_LAT)
            add(f"RC_D({j}, {g})", 0, pos[f"A_P({j}, 0)"] - LDS_LAT)
    for g in range(4):          # block 0 of the NEXT step (set 0: behind V2)
        add(f"NRC_L({g})", 0, n + pos["A_M(0)"] - LDS_LAT)
        add(f"NRC_D({g})", 0, n + pos["A_P(0, 0)"] - LDS_LAT)
    # ---- transposed fragments: ONE register set; readers in stream order: C2', C3' (qb 1 of the previous step, read at the end of the
    # previous iteration), C0, C1 (qb 0); the fragments of qb 1 of THIS step follow behind C1 (consumed by the next iteration)
    for sp in range(2):
        for nn in range(2):
            for w in "VK":
                rd = "TR_O" if w == "V" else "TR_Q"
                add(f"{rd}(0, {sp}, {nn})", pos[cname(3, w, sp, nn)] + 1, pos[cname(0, w, sp, nn)] - LDS_LAT)
                add(f"{rd}(1, {sp}, {nn})", pos[cname(1, w, sp, nn)] + 1, end)
    # ---- vector work and dS^T stores of blocks 0..2.  VE and VC both read the score set j & 1, which A(j + 2) overwrites (set 0 of
    # block 2: the next iteration's A0)
    for j in range(3):
        ready = pos[f"A_P({j}, 3)"] + LAG
        dl_scores = pos[f"A_M({j + 2})"] - LDS_LAT - 1 if j + 2 < 4 else n + pos["A_M(0)"] - LDS_LAT - 1
        for i in range(8):
            add(f"VE2({j}, {i})", ready, dl_scores)
        for i in range(8):
            sp = i >> 2
            # the packed set j & 1 is free behind its previous readers: block 0 (set 0): C2'; block 1 (set 1): C3'; block 2 (set 0): C0
            prev_reader = {0: 2, 1: 3, 2: 0}[j]
            free = max(max(pos[cname(prev_reader, "V", sp, nn)], pos[cname(prev_reader, "K", sp, nn)]) for nn in range(2)) + 1
            # next readers: C0 / C1 in this iteration; C2' in the next one
            dl = (min(pos[f"C_V({j}, {sp}, 0)"], pos[f"C_K({j}, {sp}, 0)"]) - VW) if j < 2 else end
            add(f"VC({j}, {i})", max(ready, free), min(dl, dl_scores))
        for sp in range(2):
            for t in range(2):
                add(f"DSW({j}, {sp}, {t})", ready, end)
    for k in range(16):
        add(f"DQR({k})", 0 if k < 2 else pos[f"DQM({k - 2})"] + 1, pos[f"DQM({k})"] - LDS_LAT)          # two operand register sets

    # dependencies between fillers (same or later slot, in this order inside a slot)
    after = {}
    for i in range(8):
        after[f"VCP({i})"] = [f"VEP2({i})"]
    for sp in range(2):
        for t in range(2):
            after[f"DSWP({sp}, {t})"] = [f"VCP({4 * sp + 2 * t})", f"VCP({4 * sp + 2 * t + 1})"]
    after["BARRIER()"] = [f"DSWP({sp}, {t})" for sp in range(2) for t in range(2)]
    for k in range(16):
        after[f"DQR({k})"] = ["BARRIER()"]
    for j in range(3):
        for i in range(8):
            after[f"VC({j}, {i})"] = [f"VE2({j}, {i})"]
        for sp in range(2):
            for t in range(2):
                # the dS^T image of this step is the one dQ'' (two steps back) read until the previous iteration ended: any wavefront
                # may write it only behind this iteration's barrier
                after[f"DSW({j}, {sp}, {t})"] = [f"VC({j}, {4 * sp + 2 * t})", f"VC({j}, {4 * sp + 2 * t + 1})", "BARRIER()"]
    holder = {1: [f"VEP2({i})" for i in range(8)] + [f"VCP({i})" for i in range(8)],
              2: [f"VE2(0, {i})" for i in range(8)] + [f"VC(0, {i})" for i in range(8)],
              3: [f"VE2(1, {i})" for i in range(8)] + [f"VC(1, {i})" for i in range(8)]}
    for j in (1, 2, 3):
        for g in range(4):
            after[f"RC_L({j}, {g})"] = list(holder[j]); after[f"RC_D({j}, {g})"] = list(holder[j])
    for g in range(4):
        after[f"NRC_L({g})"] = [f"VE2(2, {i})" for i in range(8)] + [f"VC(2, {i})" for i in range(8)]
        after[f"NRC_D({g})"] = list(after[f"NRC_L({g})"])
    for i in range(8):          # V1 overwrites the packed dS dwords W3' stores (set 1)
        after[f"VC(1, {i})"] = after[f"VC(1, {i})"] + [f"DSWP({i >> 2}, {(i & 3) >> 1})"]
    for i in range(8):          # V2 overwrites the packed dS dwords W0 stores (set 0)
        after[f"VC(2, {i})"] = after[f"VC(2, {i})"] + [f"DSW(0, {i >> 2}, {(i & 3) >> 1})"]
    for nm in [k for k in items if k.startswith("NR")]:
        after[nm] = list(set(after.get(nm, []) + ["BARRIER()"]))          # the next step's stage is visible behind the barrier
    for _ in range(4):          # a predecessor inherits its successors' deadlines
        for nm, preds in after.items():
            for a in preds:
                items[a][1] = min(items[a][1], items[nm][1])

    slots = [{"mfma": m, "fill": []} for m in stream] + [{"mfma": None, "fill": []}]
    placed = {}
    pending = dict(items)
    for si in range(len(slots)):
        budget = BUDGET if si < n else 10 ** 9
        for nm, ps in pinned.items():          # (pinned: first in their slot)
            if ps == si:
                slots[si]["fill"].append(nm); placed[nm] = si; budget -= COST[nm.split("(")[0]]
        while True:
            ready = [nm for nm, (e, d) in pending.items() if e <= si and all(a in placed for a in after.get(nm, []))]
            if not ready:
                break
            ready.sort(key=lambda nm: (pending[nm][1], nm))
            nm = ready[0]
            cost = COST[nm.split("(")[0]]
            urgent = pending[nm][1] <= si
            if cost > budget and not urgent:
                break
            slots[si]["fill"].append(nm)
            placed[nm] = si
            budget -= cost
            del pending[nm]
        late = [nm for nm, (e, d) in pending.items() if d <= si and si < n]
        assert not late, f"slot {si}: deadline missed for {late}"
    assert not pending, pending
    assert placed["BARRIER()"] < pinned["DMA(0)"], "the stage DMA must follow the barrier"
    return slots, pos, placed


PREV_ITEMS = ("DQM", "DQR")
PREV_LIVE_ITEMS = ("CP_V", "CP_K", "VEP2", "VCP", "DSWP")
NEXT_ITEMS = ("NRC_L", "NRC_D", "NRQB", "NRF_Q", "NRF_O")


def guard_of(item):
    name = item.split("(")[0]
    if name in PREV_ITEMS:
        return "PREV"
    if name in PREV_LIVE_ITEMS:
        return "PREV && LIVE"
    if name in NEXT_ITEMS:          # (also behind the last step: the values are then unused, and no branch merges two register classes)
        return "CUR && LIVE"
    if name in ("BARRIER", "DMA", "ST", "LD"):
        return None
    return "CUR && LIVE"


def emit(slots, path, placed_barrier):
    lines = ["// GENERATED by tools/gen_bwd1_schedule.py - do not edit.  One loop iteration of attn_bwd1p_kernel: MFMA slots and their fillers.",
             f"// {sum(1 for s in slots if s['mfma'])} matrix instructions; LAG {LAG}, LDS latency {LDS_LAT} slots, filler budget {BUDGET} cycles per slot"]
    # the iteration's vector-memory operations in issue order (an MFMA is emitted ahead of its slot's fillers)
    vm, bar, dqm0 = [], None, None
    for si, s in enumerate(slots):
        if s["mfma"] == "DQM(0)":
            dqm0 = len(vm)
        for f in s["fill"]:
            if f.split("(")[0] in ("ST", "LD", "DMA"):
                vm.append(f)
            if f == "BARRIER()":
                bar = len(vm)
    assert vm.index("DMA(4)") == len(vm) - 1 and bar is not None and dqm0 is not None and bar <= vm.index("DMA(0)")
    w1 = bar                                             # the previous iteration's DMA(4) landed: this iteration's operations stay in flight
    w2 = (len(vm) - 1 - vm.index("LD(3)")) + dqm0        # the previous iteration's LD(3) landed
    lines += [f"// vector-memory order: {' '.join(vm)}", f"#define B1_W1_YOUNGER {w1}", f"#define B1_W2_YOUNGER {w2}"]
    marks = {0: 0, placed_barrier: 1, placed_barrier + 1: 2, 24: 3, 44: 4, 64: 5}          # trace build: s_memtime at the head of these slots
    for si, s in enumerate(slots):
        parts = []
        if si in marks:
            parts.append(f"B1_TR({marks[si]});")
        if s["mfma"]:
            parts.append(f"if ({guard_of(s['mfma'])}) {{ {s['mfma']}; }}")
        for f in s["fill"]:
            g = guard_of(f)
            parts.append(f"if ({g}) {{ {f}; }}" if g else f"{f};")
        lines.append(f"/* slot {si:3d} */ " + " ".join(parts) + " B1_SB();")
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    sl, pos, placed = build()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mca-paper_amd", "csrc", "attention_bwd1_sched.inc")
    emit(sl, out, placed["BARRIER()"])
    cost = [sum(COST[f.split("(")[0]] for f in s["fill"]) for s in sl]
    print(f"{len(sl)} slots, {sum(1 for s in sl if s['mfma'])} MFMAs; filler cycles per slot: max {max(cost[:-1])}, mean {sum(cost[:-1]) / (len(cost) - 1):.1f}; "
          f"tail slot {cost[-1]}; barrier at slot {placed['BARRIER()']} -> {out}")

"""Summarises a rocprofv3 --pmc SQ-counter pass (tools/profile_round.sh, directory <dir>/sq) per kernel: counter totals over the
kernel's dispatches, the wavefront-time counters as shares of SQ_WAVE_CYCLES, and the MATRIX PIPE's busy share per SIMD cycle:
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),  kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs of the same pass
(SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed over the chip; the SQ_WAVE_* / WAIT / ACTIVE counters count quad-cycles
per wavefront, so a percentage of "wave cycles" is meaningless for it - rounds 1-4 printed 85-180 % there).
usage: sq_summary.py <dir>/sq [out.json] > profiles/rNN_attention_sq_counters.txt      (out.json: {kernel: mfma_busy}, read by bench.py)"""
import collections, csv, glob, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
N_SIMD, N_XCD = 1024, 8
print("rocprofv3 --pmc SQ counters + GRBM_GUI_ACTIVE, tools/bench_attn.py 32 (CMU structure, b = 32, H = 8, no padding), summed over each kernel's dispatches.")
print("SQ_WAVE_CYCLES / WAIT / ACTIVE are quad-cycles of wavefront time: shares of SQ_WAVE_CYCLES.  SQ_VALU_MFMA_BUSY_CYCLES is cycles per SIMD:")
print("mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) = the share of SIMD cycles the matrix pipe is executing.")
print("attn_bwd1p_kernel = one-pass backward (round 5), attn_fwd4_kernel = production forward (LDS-DMA staging), attn_fwd_kernel = its register-staged form,")
print("attn_bwd_dq / attn_bwd_dkv = the two-pass backward, attn_fwd8 / attn_bwd_dq8 / attn_bwd_dkv8 = MX-fp8 forms (BASELINE configs[4]), attn_quant_* = their quantisation passes")
busy = {}
for k in sorted(acc):
    if "SQ_WAVE_CYCLES" not in acc[k] or not k.startswith("attn"): continue
    w = acc[k]["SQ_WAVE_CYCLES"]
    print(f"{k}   ({len(disp[k])} dispatches)")
    for c, v in sorted(acc[k].items()):
        if c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES"):
            print(f"   {c:36s} {v:14.0f}")
        else:
            print(f"   {c:36s} {v:14.0f}   {v / w * 100:5.1f}% of wave cycles")
    if acc[k].get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in acc[k]:
        cyc = acc[k]["GRBM_GUI_ACTIVE"] / N_XCD
        busy[k] = acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * cyc)
        print(f"   => kernel cycles per dispatch {cyc / len(disp[k]):.0f}, mfma_busy {busy[k] * 100:.1f}% of SIMD cycles")
if len(sys.argv) > 2:
    json.dump({k: round(v, 4) for k, v in busy.items()}, open(sys.argv[2], "w"), indent=1)

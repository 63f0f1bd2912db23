"""Summarises a rocprofv3 --pmc SQ-counter pass (tools/profile_round.sh, directory <dir>/sq) per kernel: counter totals over the
kernel's dispatches and their share of SQ_WAVE_CYCLES.  usage: sq_summary.py <dir>/sq > profiles/rNN_attention_sq_counters.txt"""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
print("rocprofv3 --pmc SQ counters, tools/bench_attn.py 32 (CMU structure, b = 32, H = 8, no padding), summed over each kernel's dispatches")
print("(SQ_WAVE_CYCLES / WAIT / ACTIVE are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES is cycles: its percentage is of 4 x wave quad-cycles / 4, i.e. busy cycles per wave cycle x 4);")
print("attn_fwd4_kernel = production forward (LDS-DMA staging), attn_fwd_kernel = its register-staged form, attn_fwd8 / attn_bwd_dq8 / attn_bwd_dkv8 = MX-fp8 forms (BASELINE configs[4]), attn_quant_* = their quantisation passes")
for k in sorted(acc):
    if "SQ_WAVE_CYCLES" not in acc[k] or not k.startswith("attn"): continue
    w = acc[k]["SQ_WAVE_CYCLES"]
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:36s} {v:14.0f}   {v / w * 100:5.1f}% of wave cycles")

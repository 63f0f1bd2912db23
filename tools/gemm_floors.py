"""Every timed launch class of the b = 32 step against its two floors: algorithmic flops at the dense bf16 MFMA peak (2.5 PFLOP/s
at 2.4 GHz, and 1.65 PFLOP/s at the ~1.6 GHz the chip grants an MFMA-dense loop) and HBM bytes (counter
bytes per launch from profiles/rNN_hbm_traffic_pmc.json where the launch class maps to one kernel, else '-') at 6.2 TB/s (the
rate the LayerNorm kernels reach).  usage: gemm_floors.py profiles/r03_bench_lines.json profiles/r03_hbm_traffic_pmc.json"""
import json, sys
line = json.load(open(sys.argv[1]))[0]["line"]
pmc = json.load(open(sys.argv[2]))
kmap = {"mca_gemm_tn_acc_group": "gemm_tn_256x256_group_kernel", "mca_gemm_nt_geglu_fwd": "void gemm_nt_persist256_kernel<true>",
        "mca_gemm_nt_geglu_bwd": "void gemm_nt_persist_kernel<3, false>", "mca_gemm_nt_lnres": "void gemm_nt_256_kernel<false, 1, 1, 2>",
        "mca_attn_bwd_dkv/layer": "void attn_bwd_dkv_kernel<4>", "mca_attn_bwd_dq/layer": "void attn_bwd_dq_kernel<false, true>",
        "mca_attn_fwd/layer": "attn_fwd4_kernel", "mca_layernorm_bwd": "void ln_bwd_trunk_kernel<2>", "mca_layernorm_fwd": "void ln_fwd_trunk_kernel<2>"}
import re
print(f"{'launch class (b = 32 step)':40s} {'x/step':>6s} {'us':>7s} {'TFLOP/s':>8s} {'MFMA floor us @2.5 | @1.65 PF':>29s} {'HBM MB':>12s} {'HBM floor us':>12s} {'us / max(floor @1.65, HBM)':>26s}")
for name, v in sorted(line["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
    if v["ms_per_step"] < 0.1: continue
    us, tf = v["avg_us"], v.get("tflops", 0.0)
    gf = tf * us / 1e3          # GFLOP per launch
    f1, f2 = gf / 2.5, gf / 1.65          # us: GFLOP / (PFLOP/s)
    k = kmap.get(name)
    mb, src = ((pmc[k]["fetch_MB_x2_gfx950"] + pmc[k]["write_MB_per_launch"]), "pmc") if k in pmc else (None, "")
    m = re.match(r"mca_gemm_nt/(\d+)x(\d+)x(\d+)( f32)?( \+res)?", name)
    if mb is None and m:          # algorithmic bytes of an NT GEMM: A + B + C (+ residual)
        M, N, K = int(m.group(1)), int(m.group(2)), int(m.group(3))
        mb, src = (M * K * 2 + N * K * 2 + M * N * (4 if m.group(4) else 2) + (M * N * 4 if m.group(5) else 0)) / 1e6, "alg"
    hb = mb / 6.2 if mb else None          # MB / (TB/s) = us
    floors = [x for x in (f2 if gf > 1 else None, hb) if x]
    print(f"{name:40s} {v['launches_per_step']:6.0f} {us:7.1f} {tf:8.0f} {(f'{f1:6.1f} | {f2:6.1f}' if gf > 1 else '-'):>29s} {(f'{mb:7.0f} {src}' if mb else '-'):>12s} {(f'{hb:6.1f}' if hb else '-'):>12s} {(f'{us / max(floors):5.2f}' if floors else '-'):>26s}")

"""Timing-only ablations of the query-block forward attention kernel (attention_fwd64.hip; results are wrong, only the launch time
matters): which part of a tile's ~3,600 cycles is what.  The ablation switches are NOT in the product source: the overlay
tools/overlays/attention_fwd64_abl.patch adds them (-DF64_ABL=bits) to a copy in the variant's build directory.  Builds every
variant library first (run here, where hipcc is), then times each in its own process on the GPU box.
usage: ablate_fwd64.py build | run"""
import importlib, os, subprocess, sys
os.environ.setdefault("MCA_DEBUG", "fwd64=1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {0: "full", 128: "full, fragment registers reused inside a step", 1: "no DMA pieces in the steps", 2: "no fragment reads in the steps",
            3: "neither", 4: "no decision / slow path", 8: "no tile sync", 16: "no exponential groups", 32: "no matrix instructions",
            7: "no DMA, reads, decision", 87: "matrix instructions + sync only", 111: "exponential groups + sync only"}
def lib(bits): return os.path.join(ROOT, "mca-paper_amd", f"libmca_hip_abl{bits}.so")
if sys.argv[1] == "build":
    B = importlib.import_module("mca-paper_amd.build")
    B.build()
    for bits in VARIANTS:
        B.build_variant(lib(bits), defines=[f"F64_ABL={bits}"], only=["attention_fwd64.hip"],
                        overlays={"attention_fwd64.hip": os.path.join(ROOT, "tools", "overlays", "attention_fwd64_abl.patch")})
        print("built", lib(bits))
elif sys.argv[1] == "run":
    for bits, nm in VARIANTS.items():
        env = dict(os.environ, MCA_HIP_LIB=lib(bits))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
        print(f"F64_ABL={bits:3d} {nm:40s} {out.stdout.strip()} {out.stderr.strip()[-200:] if out.returncode else ''}", flush=True)
else:
    import torch
    P = importlib.import_module("mca-paper_amd"); H = importlib.import_module("mca-paper_amd.hip")
    res = []
    for b in (32, 8):
        cfg = P.config.cmu_model_config(batch_size=b); cfg["depth"] = 1
        torch.manual_seed(0)
        eng = P.MCA(**cfg).cuda().engine
        ws = eng.workspace(b); N, D = eng.N, eng.D
        ws["padding"].zero_()
        H.call("mca_build_keyinfo", ws["padding"].data_ptr(), eng.kgroup.data_ptr(), ws["keyinfo"].data_ptr(), ws["kflags"].data_ptr(), b, N, eng.nk_pad, H.stream_ptr())
        H.call("mca_build_keyhot", ws["keyinfo"].data_ptr(), ws["khot"].data_ptr(), b, eng.nk_pad, H.stream_ptr())
        a = ws["layers"][0]
        a["qkv"].copy_(torch.randn(b * N, 3 * D, device="cuda").bfloat16()); a["qkv"][:, :D] *= 0.18
        fn = lambda: eng._attn_fwd(a["qkv"].data_ptr(), N*3*D, 3*D, a["qkv"], D, 2*D, 3*D, a["o"], a["lse"], eng.qmask_attn, eng.sched_attn_f, ws, b, N)
        best = 1e9
        for rnd in range(3):
            for _ in range(3): fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); s.record()
            for _ in range(20): fn()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 20 * 1e3)
        res.append(f"b={b}: {best:7.1f} us")
    print("   ".join(res))

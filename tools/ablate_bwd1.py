"""Timing-only ablations of the pipelined one-pass attention backward (results of those builds are WRONG by construction): one
library per B1_ABL mask (attention_bwd1.hip), timed at b = 32 on the CMU structure with and without the kernel's memory traffic
(knob 9 bits 128 | 256 | 512).   usage: ablate_bwd1.py build|run"""
import importlib, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
sys.path.insert(0, root)
MASKS = {"base": 0, "novalu": 1, "nodq": 2, "noreads": 12, "nodsw": 16, "mfma_only": 31, "valu_lds_only": 96 + 2, "noA": 32, "noC": 64, "valu_only": 96 + 2 + 12 + 16}
if sys.argv[1] == "build":
    b = importlib.import_module("mca-paper_amd.build")
    for nm, mk in MASKS.items():
        print(b.build_variant(os.path.join(root, "mca-paper_amd", f"libabl_bwd1_{nm}.so"), [f"B1_ABL={mk}"], only=["attention_bwd1.hip"]))
else:
    for nm in MASKS:
        env = dict(os.environ, MCA_HIP_LIB=os.path.join(root, "mca-paper_amd", f"libabl_bwd1_{nm}.so"), MCA_BENCH_ATTN_ABLATE="1")
        out = subprocess.run([sys.executable, os.path.join(here, "bench_attn.py"), "32"], env=env, capture_output=True, text=True).stdout
        keep = [l for l in out.splitlines() if l.startswith("bwd one-pass ->") or "no memory" in l]
        print(f"{nm:14s}", " | ".join(l.split("layer:")[-1].split("us")[0].strip() + " us" + (" (no memory)" if "no memory" in l else "") for l in keep), flush=True)

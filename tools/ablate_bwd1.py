"""Timing-only ablations of the pipelined one-pass attention backward (results of those builds are WRONG by construction): one
library per B1_ABL mask (attention_bwd1.hip).  An ablated build computes on garbage, which changes the power the MFMAs draw and
with it the shader clock (2.25-2.4 GHz against 2.1 GHz for the product kernel): compare CYCLES, not microseconds -
tools/clocks_onepass.sh measures both (GRBM_GUI_ACTIVE / 8 XCDs and the kernel's wall time).
usage: ablate_bwd1.py build          (here; the libraries travel to the GPU box with the snapshot)
       ablate_bwd1.py run            (on the GPU box: one rocprofv3 pass per library)"""
import importlib, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
sys.path.insert(0, root)
MASKS = {"nomem": 128, "novalu": 1, "novalu_nomem": 129, "noreads": 12, "nodq": 2, "mfma_only": 31, "mfma_only_nomem": 159}
if sys.argv[1] == "build":
    b = importlib.import_module("mca-paper_amd.build")
    for nm, mk in MASKS.items():
        print(b.build_variant(os.path.join(root, "mca-paper_amd", f"libabl_bwd1_{nm}.so"), [f"B1_ABL={mk}"], only=["attention_bwd1.hip"]))
else:
    for nm in ["product"] + list(MASKS):
        env = dict(os.environ, MCA_BENCH_ATTN_ABLATE="2")
        if nm != "product":
            env["MCA_HIP_LIB"] = os.path.join(root, "mca-paper_amd", f"libabl_bwd1_{nm}.so")
        out = subprocess.run(["bash", os.path.join(here, "clocks_onepass.sh"), os.path.join(root, "gpurun_out", f"clk_{nm}")], env=env, capture_output=True, text=True, cwd=root).stdout
        print(f"{nm:16s}", (out.strip().splitlines() or ["(no output)"])[-1], flush=True)

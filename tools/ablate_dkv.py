"""Timing-only ablations of the dkv pass of the attention backward (results of those builds are WRONG by construction):
builds one library per DKV_ABL mask and times the layer backward at b = 32 on the CMU structure.  The switches are not in the
product source: tools/overlays/attention_bwd2_dkv_abl.patch adds them to a copy in the variant's build directory.
usage: ablate_dkv.py build|run"""
import importlib, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.dirname(here)
sys.path.insert(0, root)
MASKS = {"base": 0, "noexp": 1, "noprod": 2, "notr": 4, "noexp_notr": 5, "noprod_noexp": 3}
if sys.argv[1] == "build":
    b = importlib.import_module("mca-paper_amd.build")
    for nm, mk in MASKS.items():
        print(b.build_variant(os.path.join(root, "mca-paper_amd", f"libabl_dkv_{nm}.so"), [f"DKV_ABL={mk}"], only=["attention_bwd2.hip"],
                              overlays={"attention_bwd2.hip": os.path.join(here, "overlays", "attention_bwd2_dkv_abl.patch")}))
else:
    for nm in MASKS:
        env = dict(os.environ, MCA_HIP_LIB=os.path.join(root, "mca-paper_amd", f"libabl_dkv_{nm}.so"), MCA_BENCH_ATTN_ONLY="bwd")
        out = subprocess.run([sys.executable, os.path.join(here, "bench_attn.py"), "32"], env=env, capture_output=True, text=True).stdout
        print(f"{nm:14s}", out.strip().split("->")[-1][:70], flush=True)

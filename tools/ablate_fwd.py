"""Timing-only ablations of the forward attention kernel (second form): builds one library per FW_ABL mask (results of those
builds are WRONG by construction) and times the layer forward at b = 32 on the CMU structure.  usage: ablate_fwd.py build|run"""
import importlib, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
VARIANTS = {"base": [], "noexp": ["FW_ABL=1"], "adds": ["FW_ABL=2"], "nosum": ["FW_ABL=64"], "noPV": ["FW_ABL=4"], "noS": ["FW_ABL=8"],
            "nostage": ["FW_ABL=16"], "nobarrier": ["FW_ABL=32"], "noexp_nosum": ["FW_ABL=65"], "w3": ["FW2_MINWAVES=3"],
            "w3_adds": ["FW2_MINWAVES=3", "FW_ABL=2"]}
if sys.argv[1] == "build":
    b = importlib.import_module("mca-paper_amd.build")
    b.build()
    for name, defs in VARIANTS.items():
        b.build_variant(os.path.join(root, "mca-paper_amd", f"libabl_{name}.so"), defs, only=("attention_fwd.hip",))
        print("built", name, flush=True)
else:
    for rep in range(2):
        for name in VARIANTS:
            env = dict(os.environ, MCA_HIP_LIB=os.path.join(root, "mca-paper_amd", f"libabl_{name}.so"), MCA_BENCH_ATTN_ONLY="fwd")
            out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_attn.py"), "32"], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if "->" in l]
            print(f"{name:14s} {line[0].split(':')[1].split('us')[0].strip() if line else out.stderr[-300:]} us", flush=True)

"""Builds libmca_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is plain HIP with a
C ABI (include/mca_hip.h) and is loaded with ctypes."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmca_hip.so")
SOURCES = ["elementwise.hip", "gemm.hip", "attention_fwd.hip", "attention_bwd2.hip", "attention_bwd1.hip", "attention_fp8.hip", "loss.hip", "optim.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++20", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
# per-file extras: keep the attention accumulators in VGPRs (the softmax VALU works on them in place; the default
# AGPR form costs 256 v_accvgpr moves per key tile)
EXTRA = {"attention_fwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "attention_bwd2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         "attention_fp8.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
         # hand-placed issue order: the SLP vectoriser packs adjacent f32 adds into v_pk_add_f32 and moves whole groups with them
         # generated issue order (attention_bwd1_sched.inc): keep single-instruction multiplies single
         "attention_bwd1.hip": ["-fno-slp-vectorize"]}


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mca_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


TRACE_OUT = os.path.join(HERE, "libmca_hip_trace.so")


def build_variant(out_path: str, defines=(), only=None, overlays=None) -> str:
    """An A/B build of the same ABI with extra -D defines (tools/ab_lib_*.py load it through MCA_HIP_LIB).  only: the sources the
    defines affect; every other object is taken from the product build directory (run build() first).
    overlays: {source name: patch file}: measurement code that does NOT live in the product sources (timing-only ablations,
    tools/overlays/*.patch) is patched into a copy of the source in the variant's build directory first; a patch that no longer
    applies fails the build."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(HERE, "build_" + os.path.splitext(os.path.basename(out_path))[0])
    os.makedirs(bdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        if only is not None and src not in only:
            objs.append(os.path.join(HERE, "build", src.replace(".hip", ".o")))
            continue
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        src_path = os.path.join(CSRC, src)
        if overlays and src in overlays:
            src_path = os.path.join(bdir, src)
            r = subprocess.run(["patch", "-s", "-o", src_path, os.path.join(CSRC, src), overlays[src]], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"overlay {overlays[src]} does not apply to {src}:\n{r.stdout}")
        cmd = [hipcc, *FLAGS, *[f"-D{d}" for d in defines], *EXTRA.get(src, []), f"-I{CSRC}", "-c", src_path, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out_path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return out_path


def build(force: bool = False, verbose: bool = True, trace: bool = False) -> str:
    """trace=True: a second library (libmca_hip_trace.so, same ABI) whose attention kernels carry the s_memtime stamps read by
    tools/trace_attn_*.py (load it with MCA_HIP_LIB); the stamps pin the instruction order, so the product build has none."""
    out_path = TRACE_OUT if trace else OUT
    if not trace and not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build_trace" if trace else "build")
    os.makedirs(bdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *(["-DMCA_TRACE_BUILD"] if trace else []), *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out_path]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if verbose:
        print(f"built {out_path}")
    return out_path


if __name__ == "__main__":
    build(force="--force" in sys.argv, trace="--trace" in sys.argv)

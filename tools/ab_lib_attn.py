"""Attention kernel times of two builds of the library, alternating processes.  usage: ab_lib_attn.py <libA.so> <libB.so> [rounds]"""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
libs = sys.argv[1:3]; rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, MCA_HIP_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(here, "bench_attn.py")], env=env, capture_output=True, text=True)
        lines = [x for x in out.stdout.splitlines() if "->" in x]
        print(os.path.basename(l), " | ".join(x.split(":")[0].split("->")[0].strip() + " " + x.split(":")[1].split("us")[0].strip() + "us" for x in lines), flush=True)
        if out.returncode: print(out.stderr[-2000:])

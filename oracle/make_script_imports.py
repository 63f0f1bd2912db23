"""Test infrastructure (not product code): writes tests/golden/ref_script_imports.json, the list of names the reference's entry
scripts import from the repository's own modules (``from model import MCA, EAO`` ...: train_accel_gpu.py:12-17,
infer_accel_gpu.py:12-17; the linear-probe script is out of scope).  The names are data (a module -> names table); no source text is kept.
tests/test_host_cpu.py imports exactly these names from this repository's drop-in modules.

    python oracle/make_script_imports.py          # only in the container that has /root/reference
"""
import ast
import json
import os

REF = "/root/reference"
LOCAL = ("model", "encoders", "utils")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_script_imports.json")


def main():
    table = {}
    for script in ("train_accel_gpu.py", "infer_accel_gpu.py"):
        tree = ast.parse(open(os.path.join(REF, script)).read())
        mods = {}
        for node in tree.body:
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] in LOCAL:
                mods.setdefault(node.module, []).extend(a.name for a in node.names)
        table[script] = mods
    json.dump(table, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()

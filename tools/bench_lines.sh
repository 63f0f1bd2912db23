#!/bin/bash
# The bench lines recorded per round under profiles/rNN_bench_lines.json (run on the GPU box from the repo root:
# bash tools/bench_lines.sh <outdir>); writes one JSON per command and <outdir>/lines.json.
set -o pipefail
OUT="${1:-gpurun_out/lines}"; mkdir -p "$OUT"
run() { local tag="$1"; shift; echo "== $tag: $*"; timeout -k 10 500 "$@" > "$OUT/$tag.json" 2> "$OUT/$tag.err" || { echo "FAILED $tag"; tail -3 "$OUT/$tag.err"; return 1; }; cut -c1-170 "$OUT/$tag.json" | tail -1; }
run b32_default python bench.py &&
run b32_nosample python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run b32_host python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 --inputs host &&
run b32_ragged python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 --lengths uniform --p-drop 0.2 &&
run b8_nosample python bench.py --batch 8 --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run b8_mma python bench.py --batch 8 --variant mma --p-drop 0.4 --steps 60 --warmup 5 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run eao_b8 python bench.py --variant eao --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run long_bf16 python bench.py --workload long --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run long_fp8 python bench.py --workload long --attn fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run long_bf16_again python bench.py --workload long --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run long_fp8_again python bench.py --workload long --attn fp8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0
python3 - "$OUT" <<'PY'
import glob, json, os, sys
out = []
for f in sorted(glob.glob(sys.argv[1] + "/*.json"), key=os.path.getmtime):
    if f.endswith("lines.json"): continue
    try: out.append({"what": os.path.basename(f)[:-5], "line": json.loads(open(f).read().strip().splitlines()[-1])})
    except Exception as e: out.append({"what": os.path.basename(f)[:-5], "error": str(e)})
json.dump(out, open(sys.argv[1] + "/lines.json", "w"), indent=1)
PY

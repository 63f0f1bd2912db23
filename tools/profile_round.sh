#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_* (run on the GPU box from the repo root: bash tools/profile_round.sh <outdir>).
# Kernel durations and counters come from SEPARATE runs (never --pmc together with --stats); weight gradients stay on the main
# stream (MCA_DEBUG=overlap_wgrad=0) so that every duration is a kernel's own.
set -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/prof}"); mkdir -p "$OUT"
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
export MCA_DEBUG=overlap_wgrad=0 PYTHONPATH=$ROOT
B="$ROOT/bench.py"
run() { echo "== $*"; timeout -k 10 400 "$@" > "$OUT/last.log" 2>&1 || { echo "FAILED: $*"; tail -5 "$OUT/last.log"; return 1; }; }
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o t -- python3 $B --launch eager --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -o t -- python3 $B --launch eager --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -o t -- python3 $B --launch eager --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_long" -o t -- python3 $B --launch eager --workload long --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_long_fp8" -o t -- python3 $B --launch eager --workload long --attn fp8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --sustain-seconds 0 &&
run rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/sq" -o t -- python3 $ROOT/tools/bench_attn.py 32
cd "$ROOT"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_same_box.json" 2> "$OUT/bench_same_box.err"
# keep only the summaries (the traces are tens of MB)
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
du -sh "$OUT"

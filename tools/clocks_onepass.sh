#!/bin/bash
# Effective shader clock and cycle count of the one-pass attention backward under its timing-only ablations (knob 9 bits): the
# ablated variants compute on garbage, which changes the power the MFMAs draw and with it the clock - microseconds of such a
# variant are not comparable with the product's, cycles are.  Run on the GPU box from the repo root; writes <out>/clocks.txt.
set -o pipefail
ROOT=$(pwd); OUT=$(realpath -m "${1:-gpurun_out/clk1}"); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT MCA_BENCH_ATTN_ABLATE=${MCA_BENCH_ATTN_ABLATE:-1}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d $OUT/p -o t -- python3 $ROOT/tools/bench_attn.py 32 > $OUT/run.log 2>&1 || { echo FAILED; tail -5 $OUT/run.log; exit 1; }
python3 - $OUT/p <<'PY' > $OUT/clocks.txt
import csv, glob, sys
dur = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur and "attn_bwd1" in dur[r["Dispatch_Id"]][0]:
            rows.append((int(r["Dispatch_Id"]), dur[r["Dispatch_Id"]][0], float(r["Counter_Value"]), dur[r["Dispatch_Id"]][1]))
rows.sort()
print("case (22 dispatches each, the last 10 averaged)   kernel   us   Mcycles per XCD   GHz")
for i in range(0, len(rows), 22):
    ch = rows[i:i + 22][-10:]
    us = sum(c[3] for c in ch) / len(ch) / 1e3; cyc = sum(c[2] for c in ch) / len(ch) / 8
    print(f"case {i // 22:2d}  {ch[0][1]:20s} {us:8.1f} us  {cyc / 1e6:7.3f} Mcycles  {cyc / (us * 1e3):5.2f} GHz")
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
cat $OUT/clocks.txt

#!/bin/bash
# SQ / LDS counters of the grouped weight-gradient GEMM at the step's shapes (two rocprofv3 --pmc passes of
# tools/ablate_tn_group.py, summed by tools/pmc_sum.py).  Run on the GPU box from the repo root: bash tools/pmc_tn_group.sh [outdir]
set -o pipefail
ROOT=$(pwd); OUT=$(realpath -m "${1:-gpurun_out/pmc_tn}"); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT
for v in "whole kernel"; do
  tag=$(echo "$v" | tr -c 'a-zA-Z\n' '_')
  MCA_ABLATE_ONLY="$v" timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_$tag -o t -- python3 $ROOT/tools/ablate_tn_group.py 32 > $OUT/pmc_$tag.log 2>&1 || { echo FAILED $v; tail -5 $OUT/pmc_$tag.log; exit 1; }
  echo "== $v"; python3 $ROOT/tools/pmc_sum.py $OUT/pmc_$tag tn_256x256_group
  MCA_ABLATE_ONLY="$v" timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC -d $OUT/pmc2_$tag -o t -- python3 $ROOT/tools/ablate_tn_group.py 32 > $OUT/pmc2_$tag.log 2>&1 || { echo FAILED2 $v; tail -5 $OUT/pmc2_$tag.log; exit 1; }
  python3 $ROOT/tools/pmc_sum.py $OUT/pmc2_$tag tn_256x256_group
  find $OUT -name "*kernel_trace.csv" -delete
done

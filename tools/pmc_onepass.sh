#!/bin/bash
# Counter passes on the one-pass attention backward (tools/bench_attn.py 32, only the "bwd one-pass" case): dynamic instruction
# counts, where wavefront cycles go, LDS conflicts, fabric traffic.  Run on the GPU box from the repo root; sums per dispatch in
# <out>/*.sum (tools/pmc_sum.py).  Each pass is its own run (never --pmc together with --stats).
set -o pipefail
ROOT=$(pwd); OUT=$(realpath -m "${1:-gpurun_out/pmc1}"); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT MCA_BENCH_ATTN_ONLY="bwd one-pass"
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_IFETCH SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL" \
         "${PMC_EXTRA:-FETCH_SIZE}" "${PMC_EXTRA2:-WRITE_SIZE}"; do
  i=$((i+1)); d=$OUT/pass$i
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $d -o t -- python3 $ROOT/tools/bench_attn.py 32 > $d.log 2>&1 || echo "FAILED $c"
  python3 $ROOT/tools/pmc_sum.py $d attn_bwd1 > $d.sum 2>&1 || true
  find $d -name "*kernel_trace.csv" -delete; find $d -name "*counter_collection.csv" -size +4M -delete
done
cat $OUT/*.sum

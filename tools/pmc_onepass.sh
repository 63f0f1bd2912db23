set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$ROOT MCA_BENCH_ATTN_ONLY="bwd one-pass"
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
for c in FETCH_SIZE WRITE_SIZE "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_GATE_EN1_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  d=$OUT/$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $d -o t -- python3 $ROOT/tools/bench_attn.py 32 > $d.log 2>&1 || echo "FAILED $c"
  python3 $ROOT/tools/pmc_sum.py $d onepass > $d.sum 2>&1 || true
  find $d -name "*kernel_trace.csv" -delete
done
tail -n 20 $OUT/*.sum

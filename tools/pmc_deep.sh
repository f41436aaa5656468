#!/bin/bash
# Deeper PMC passes (memory pipeline) for one bench workload; run on the GPU box.
# Each pass is time-boxed: an over-subscribed counter group makes rocprofv3 abort and then hang.
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for c in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
         "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
         "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum"; do
  name=$(echo $c | tr ' ' '+' | cut -c1-60)
  echo "pass $c" >> $out/progress.log
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_$name.log 2>&1 || echo "pmc pass $c failed" >> $out/errors.log
done

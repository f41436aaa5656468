#!/bin/bash
# PMC passes for the REDUCE compaction kernels (tools/reduce_prof.py); run on the GPU box.
out=gpurun_out/prof_reduce_pmc
mkdir -p $out
export TMPDIR=/tmp
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "TA_TA_BUSY_sum TD_TD_BUSY_sum" "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $c | tr ' ' '+' | cut -c1-60)
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$name -- python3 tools/reduce_prof.py unique > $out/pmc_$name.log 2>&1 || echo "pmc pass $c failed" >> $out/errors.log
done

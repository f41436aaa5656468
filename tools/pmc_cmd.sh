#!/bin/bash
# usage: tools/pmc_cmd.sh <tag> <kernel-name-pattern> <script.py> [args...] -- rocprofv3 kernel trace + a handful of PMC passes (each in its own run)
# for any python script of tools/ (the program after `--` is python3 itself: no wrapper between the profiler and the GPU process)
tag=$1; pat=$2; shift 2
out=gpurun_out/pc_$tag
mkdir -p $out; export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "$@" > $out/trace.log 2>&1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU"; do
  name=$(echo $c | tr ' ' '+' | cut -c1-50)
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$name -- python3 "$@" > $out/pmc_$name.log 2>&1 || echo "pmc pass $c failed" >> $out/errors.log
done
python3 tools/summarize_prof.py $out $out/summary > /dev/null 2>&1
grep -A40 "$pat" $out/summary.txt | head -${LINES_OUT:-70}

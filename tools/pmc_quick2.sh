#!/bin/bash
# usage: tools/pmc_quick2.sh <tag> <bench args...>  -- a handful of rocprofv3 PMC passes (+ kernel trace) for one bench workload
tag=$1; shift
out=gpurun_out/pq_$tag
mkdir -p $out; export TMPDIR=/tmp
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra "$@" > $out/trace.log 2>&1
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  name=$(echo $c | tr ' ' '+' | cut -c1-50)
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $out/pmc_$name.log 2>&1 || echo "pmc pass $c failed" >> $out/errors.log
done
python3 tools/summarize_prof.py $out $out/summary > /dev/null 2>&1
grep -A40 "comb_hij_plan_kernel\|filtered_kernel\|eloc_rbm_kernel\|eloc_sample_space_keys_kernel\|reduce_onepass\|reduce_contract" $out/summary.txt | head -${LINES_OUT:-120}

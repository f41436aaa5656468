#!/bin/bash
# usage: tools/pmc_tcc.sh <tag> <bench args...> -- L2 (TCC) and LDS counters of one bench workload, one rocprofv3 pass per group
tag=$1; shift
out=gpurun_out/tcc_$tag
mkdir -p $out; export TMPDIR=/tmp
for c in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_ATOMIC_sum" "TCC_EA_ATOMIC_sum TCC_EA_RD_UNCACHED_32B_sum TCC_READ_sum TCC_WRITE_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
         "TCC_TAG_STALL_sum TCC_EA_RDREQ_LEVEL_sum TCC_BUSY_sum"; do
  name=$(echo $c | tr ' ' '+' | cut -c1-50)
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $out/pmc_$name.log 2>&1 || echo "pmc pass $c failed" >> $out/errors.log
done
python3 tools/summarize_prof.py $out $out/summary > /dev/null 2>&1
grep -A30 "== rocprofv3 --pmc" $out/summary.txt | grep -A28 "reduce_onepass_list_kernel" | head -40
cat $out/errors.log 2>/dev/null

#!/bin/bash
# rocprofv3 passes for the fused RBM local-energy kernel (run on the GPU box): trace + a few PMC groups.
out=gpurun_out/prof_rbm
mkdir -p $out
export TMPDIR=/tmp
W="--workload fe2s2_eloc_rbm --no-cpu-baseline --no-extra"
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 3 $W > $out/trace.log 2>&1
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  name=$(echo $c | tr ' ' '+' | cut -c1-60)
  echo "pass $c" >> $out/progress.log
  timeout -k 5 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 $W > $out/pmc_$name.log 2>&1 || echo "pmc pass $c failed" >> $out/errors.log
done

#!/bin/bash
# Re-profile every workload whose profile bench.py's `roofline` quotes (GPU box, from the repo root): tools/pmc_roofline.py per workload,
# outputs copied to gpurun_out/profiles_out/ (gpurun merges only gpurun_out/ back): copy them into profiles/ afterwards.
#   usage: tools/pmc_all.sh [tag-prefix, default r04]
pre=${1:-r04}
mkdir -p gpurun_out/profiles_out
run() {  # workload, pmc name, walkers (0 = default), extra args
  local w=$1 name=$2 nw=$3; shift 3
  timeout -k 10 500 python3 tools/pmc_roofline.py --workload $w --pmc-name $name --tag ${pre}_$name $([ "$nw" != 0 ] && echo --walkers $nw) --steps 20 "$@" > gpurun_out/profiles_out/${pre}_$name.log 2>&1
  echo "$w -> $name: rc $?"
  cp profiles/${pre}_$name.txt profiles/pmc_$name.json gpurun_out/profiles_out/ 2>/dev/null
}
run fe2s2_reduce_vmc_step fe2s2_reduce_vmc_step 0
run fe2s2_eloc_sample_space fe2s2_eloc_sample_space 0
run fe2s2_dropin fe2s2_dropin 0
run fe2s2_eloc_rbm fe2s2_eloc_rbm 0
run syn120_dropin syn120_dropin 64
run syn184_dropin syn184_dropin 16
run syn120_eloc_sample_space syn120_eloc_sample_space_indexed 0
run syn184_eloc_sample_space syn184_eloc_sample_space_indexed 0
run syn56_reduce_vmc_step syn56_reduce_vmc_step 4096 --kernel reduce_onepass_list_flush_kernel
run syn120_reduce_vmc_step syn120_reduce_vmc_step 0 --kernel reduce_onepass_list_flush_kernel
run syn184_reduce_vmc_step syn184_reduce_vmc_step 4096 --kernel reduce_onepass_list_flush_kernel
ls gpurun_out/profiles_out

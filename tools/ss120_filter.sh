#!/bin/bash
# sample-space kernel at sorb 120 (64 Ki keys): LDS filter size against time
for cfg in "0 8" "131072 8" "262144 8" "524288 16"; do
  set -- $cfg
  echo "PYNQS_FILTER_BITS=$1 PYNQS_FILTER_PER_KEY=$2"
  PYNQS_FILTER_BITS=$1 PYNQS_FILTER_PER_KEY=$2 timeout -k 10 120 python bench.py --workload ${WL:-syn120_eloc_sample_space} --walkers ${NW:-2048} --steps ${STEPS:-5} --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'])" || exit 1
done

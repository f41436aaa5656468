#!/bin/bash
# sample-space kernel at sorb 120: chunks per walker / XCD-aware order against time
for cfg in "4096 0" "32768 1" "131072 0" "131072 1" "524288 1"; do
  set -- $cfg
  echo "PYNQS_WANT_WG=$1 PYNQS_XCD_MAP=$2"
  PYNQS_FILTER_BITS=262144 PYNQS_WANT_WG=$1 PYNQS_XCD_MAP=$2 timeout -k 10 120 python bench.py --workload ${WL:-syn120_eloc_sample_space} --walkers ${NW:-2048} --steps ${STEPS:-5} --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], j['parity'])" || exit 1
done

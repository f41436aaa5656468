#!/bin/bash
# sample-space kernel with a table too large for an LDS filter (1 M keys): global filter alone vs no filter
for fb in 262144 0; do
  echo "PYNQS_FILTER_BITS=$fb"
  PYNQS_FILTER_BITS=$fb timeout -k 10 200 python bench.py --workload syn120_eloc_sample_space --walkers 1024 --keys 1000000 --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], j['parity'])" || exit 1
done

#!/bin/bash
# sample-space kernel timings (Fe2S2 8192 walkers, sorb 120 2048 walkers) + the tests that cover it
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "sample_space or energy or fuzz" > gpurun_out/t.log 2>&1; tail -n 2 gpurun_out/t.log
for cfg in "fe2s2_eloc_sample_space 8192 1000 100" "syn120_eloc_sample_space 2048 5 1"; do
  set -- $cfg
  timeout -k 10 100 python bench.py --workload $1 --walkers $2 --steps $3 --warmup $4 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['ms_per_step'], j['roofline']['kernel_ms'], j['parity'])"
done

#!/bin/bash
# Interleaved A/B of prebuilt library variants in build_ab/ (same box, alternating runs).
for round in 1 2 3; do
  for v in "$@"; do
    PYNQS_AMD_LIB=$PWD/build_ab/lib_$v.so python bench.py --no-cpu-baseline --no-extra --steps 40 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['roofline']['kernel_ms'],4), d['parity']['max_abs_diff_vs_oracle'])"
  done
done

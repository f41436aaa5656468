#!/bin/bash
# Interleaved A/B of prebuilt library variants in build_ab/ (same box, alternating runs).
#   tools/ab.sh "<bench args>" <variant> <variant> ...      e.g. tools/ab.sh "--workload fe2s2_dropin" p4nb p6nb
args=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    PYNQS_AMD_LIB=$PWD/build_ab/lib_$v.so python bench.py --no-cpu-baseline --no-extra --steps 300 --warmup 20 $args 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', round(d['roofline']['kernel_ms'],4), d['parity'])"
  done
done

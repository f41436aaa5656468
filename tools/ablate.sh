#!/bin/bash
# Diagnostic builds of the library with parts of the Dab loop removed (run on the GPU box).
set -e
mkdir -p gpurun_out/abl
for v in BASE NOGATHER NOSTORE; do
  flag=""; [ $v != BASE ] && flag="-DPYNQS_ABL_$v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flag -o gpurun_out/abl/lib_$v.so pynqs_amd/csrc/*.hip
done
for v in BASE NOGATHER NOSTORE; do
  for extra in "" "--no-comb"; do
    PYNQS_AMD_LIB=$PWD/gpurun_out/abl/lib_$v.so python bench.py --no-cpu-baseline --steps 30 $extra "$@" 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', '$extra', 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['parity'])"
  done
done

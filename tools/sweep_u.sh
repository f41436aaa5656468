#!/bin/bash
# Diagnostic: sweep the number of excitations in flight per lane (run on the GPU box).
set -e
mkdir -p gpurun_out/abl
for u in 1 2 4 8; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPYNQS_U=$u -o gpurun_out/abl/lib_U$u.so pynqs_amd/csrc/*.hip
  for extra in "" "--no-comb"; do
  PYNQS_AMD_LIB=$PWD/gpurun_out/abl/lib_U$u.so python bench.py --no-cpu-baseline --steps 30 $extra "$@" 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('U=$u', '$extra', 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['parity'])"
  done
done

#!/bin/bash
# Diagnostic: sweep pair slots in flight per lane (PYNQS_U) and the LDS staging tile (run on the GPU box).
mkdir -p gpurun_out/abl
for cfg in "1 2048" "2 2048" "4 2048" "2 1024" "2 512" "1 1024"; do
  set -- $cfg; u=$1; t=$2
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DPYNQS_U=$u -DPYNQS_DIAG_TILE=$t -o gpurun_out/abl/lib_U${u}_T$t.so pynqs_amd/csrc/*.hip || continue
  for wl in "fe2s2_dropin 8192" "syn120_dropin 64"; do
    set -- $wl
    PYNQS_AMD_LIB=$PWD/gpurun_out/abl/lib_U${u}_T$t.so python bench.py --no-cpu-baseline --no-extra --steps 30 --workload $1 --walkers $2 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('U=$u tile=$t', '$1', 'kernel_ms', round(d['roofline']['kernel_ms'],4), d['parity']['max_abs_diff_vs_oracle'])"
  done
done

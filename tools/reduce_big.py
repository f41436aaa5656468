"""Compaction throughput (count + emit, |H| >= eps) at the BASELINE sizes with synthetic dense integrals."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pynqs_amd import energy as E

sorb, no, nw = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
eps = float(sys.argv[4]) if len(sys.argv) > 4 else 0.495
dev = torch.device("cuda")
h1, h2 = bench.synth_integrals(sorb)
h1, h2 = h1.to(dev), h2.to(dev)
x = bench.synth_walkers(nw, sorb, no, no, 4321).to(dev)
for _ in range(2):
    r = E.reduce_compact(x, h1, h2, sorb, 2 * no, no, no, eps)
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    r = E.reduce_compact(x, h1, h2, sorb, 2 * no, no, no, eps)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
ncomb = int(bench.algorithmic_bytes_dropin(sorb, 2 * no, no, no)[1])
print(f"sorb {sorb} ncomb {ncomb} walkers {nw} eps {eps}: {ms:.3f} ms -> {nw / ms * 1e3:.3e} walkers/s, {nw * ncomb / ms / 1e6:.1f} G columns/s, kept {r[1].numel() / nw:.0f} per walker")

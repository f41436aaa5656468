"""Soak test of the REDUCE compaction (count + emit) and the semi-stochastic draws: repeated calls must return
identical records (no atomics anywhere) and never stall."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import energy
d = np.load("tests/golden/fe2s2_inputs.npz")
dev = torch.device("cuda")
n, reps = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 300
x = torch.from_numpy(d["ci_space"][:n].copy()).to(dev)
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
ref = energy.reduce_compact(x, h1e, h2e, 40, 30, 15, 15, 1e-2)
t0 = time.perf_counter()
for i in range(reps):
    out = energy.reduce_compact(x, h1e, h2e, 40, 30, 15, 15, 1e-2)
    assert all(torch.equal(a, b) for a, b in zip(ref, out)), i
torch.manual_seed(0)
ref2 = energy.reduce_compact_sampled(x[:1024], h1e, h2e, 40, 30, 15, 15, 1e-2, 1000, seed=3)
for i in range(reps // 10):
    torch.manual_seed(0)
    out2 = energy.reduce_compact_sampled(x[:1024], h1e, h2e, 40, 30, 15, 15, 1e-2, 1000, seed=3)
    assert all(torch.equal(a, b) for a, b in zip(ref2[1], out2[1])), i
torch.cuda.synchronize()
print(f"ok: {reps} compactions + {reps // 10} semi-stochastic selections identical, {time.perf_counter() - t0:.1f} s")

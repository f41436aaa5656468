"""Timing of the fused RBM local energy (pynqs_eloc_rbm) on Fe2S2 walkers; HIP events on the launch stream."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx

d = np.load("tests/golden/fe2s2_inputs.npz")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 2
kind = sys.argv[3] if len(sys.argv) > 3 else "real"  # real / tanh / pRBM
sorb = 40
x = torch.from_numpy(d["ci_space"][:n].copy()).cuda()
h1e, h2e = torch.from_numpy(d["h1e"]).cuda(), torch.from_numpy(d["h2e"]).cuda()
g = torch.Generator().manual_seed(7)
H = int(alpha * sorb)
W = (0.01 * (torch.rand(H, sorb, generator=g, dtype=torch.float64) - 0.5)).cuda()
hb = (0.01 * (torch.rand(H, generator=g, dtype=torch.float64) - 0.5)).cuda()
vb = (1.0 * (torch.rand(sorb, generator=g, dtype=torch.float64) - 0.5)).cuda()
tab = cx.RBMTable(W, hb, vb)
for _ in range(3):
    e, p = cx.eloc_rbm(x, h1e, h2e, tab, sorb, 30, 15, 15, rbm_type=kind)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 20
a.record()
for _ in range(K):
    e, p = cx.eloc_rbm(x, h1e, h2e, tab, sorb, 30, 15, 15, rbm_type=kind)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / K
ncomb = 7876
print(f"n={n} H={H} {kind}: {ms:.3f} ms/launch, {n/ms*1e3:.3e} E_loc/s, {n*ncomb*H*2/ms/1e9:.2f} TFLOP/s-equivalent (2 f64 ops per exc*h), mean {complex(e.mean()):.6f}")

"""Timing of the fused complex-RBM local energy (pynqs_eloc_crbm) on Fe2S2 walkers.  usage: python tools/crbm_time.py [n] [alpha]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx
d = np.load("tests/golden/fe2s2_inputs.npz")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 1
sorb = 40
x = torch.from_numpy(d["ci_space"][:n].copy()).cuda()
h1e, h2e = torch.from_numpy(d["h1e"]).cuda(), torch.from_numpy(d["h2e"]).cuda()
g = torch.Generator().manual_seed(7)
H = int(alpha * sorb)
W = (0.02 * (torch.rand(H, sorb, 2, generator=g, dtype=torch.float64) - 0.5)).cuda()
hb = (0.02 * (torch.rand(H, 2, generator=g, dtype=torch.float64) - 0.5)).cuda()
vb = (0.05 * (torch.rand(sorb, 2, generator=g, dtype=torch.float64) - 0.5)).cuda()
tab = cx.CRBMTable(W, hb, vb)
for _ in range(3):
    e, p = cx.eloc_crbm(x, h1e, h2e, tab, sorb, 30, 15, 15)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 10
a.record()
for _ in range(K):
    e, p = cx.eloc_crbm(x, h1e, h2e, tab, sorb, 30, 15, 15)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / K
print(f"n={n} H={H} complex: {ms:.3f} ms/launch, {n/ms*1e3:.3e} E_loc/s, {n*7876*H*11/ms/1e9:.2f} T f64 instr-lanes/s (11 per exc*h), mean {complex(e.mean()):.6f}")

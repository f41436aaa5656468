"""The Fe2S2 example's eloc_param (REDUCE, use_unique, use_LUT, eps 1e-2, eps_sample 1000) on 8192 walkers: time per call
with and without the look-up table of the sampled determinants (psi of the table = the ansatz' own values)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import energy as E, public_function as pf
from pynqs_amd.rbm import RealRBM
torch.set_default_dtype(torch.float64)
d = np.load("tests/golden/fe2s2_inputs.npz")
n, sorb, nele, noA, noB = 8192, 40, 30, 15, 15
dev = torch.device("cuda")
x = torch.from_numpy(d["ci_space"][:n].copy()).to(dev)
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
g = torch.Generator().manual_seed(7)
rbm = RealRBM(0.01 * (torch.rand(2 * sorb, sorb, generator=g) - 0.5), 0.01 * (torch.rand(2 * sorb, generator=g) - 0.5), 0.1 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
ab = lambda xx, func: pf.ansatz_batch(func, xx, 150_000, sorb, dev, torch.double)
keys = torch.from_numpy(d["ci_space"].copy()).to(dev)
with torch.no_grad():
    lut = pf.WavefunctionLUT(keys, ab(keys, rbm), sorb, device=dev)
E.FUSED_RBM = False
for name, kw in (("no LUT", {}), ("LUT of 18496 determinants", {"WF_LUT": lut})):
    for eps_sample in (0, 1000):
        f = lambda: E.local_energy(x, h1e, h2e, rbm, ab, sorb, nele, noA, noB, reduce_psi=True, eps=1e-2, eps_sample=eps_sample, use_unique=True, **kw)
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            e = f()[0]
        torch.cuda.synchronize()
        print(f"{name:28s} eps_sample={eps_sample:5d}: {(time.perf_counter() - t0) / 5 * 1e3:7.3f} ms   <E> = {e.mean().item():.6f}")

import os, sys
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
ab = lambda xx, func: pf.ansatz_batch(func, xx, 2_000_000, sorb, dev, torch.double)
for _ in range(4):
    E.local_energy(x, h1e, h2e, rbm, ab, sorb, nele, noA, noB, reduce_psi=True, eps=1e-2, eps_sample=1000, use_unique=True)
torch.cuda.synchronize()

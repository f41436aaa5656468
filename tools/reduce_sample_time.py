"""The Fe2S2 example's production setting: REDUCE with eps = 1e-2 AND eps_sample = 1000 (semi-stochastic)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import energy as E, public_function as pf
from pynqs_amd.rbm import RealRBM
torch.set_default_dtype(torch.float64)
d = np.load("tests/golden/fe2s2_inputs.npz")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
sorb, nele, noA, noB = 40, 30, 15, 15
dev = torch.device("cuda")
x = torch.from_numpy(d["ci_space"][:n].copy()).to(dev)
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
g = torch.Generator().manual_seed(7)
rbm = RealRBM(0.01 * (torch.rand(2 * sorb, sorb, generator=g) - 0.5), 0.01 * (torch.rand(2 * sorb, generator=g) - 0.5), 0.1 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
ab = lambda xx, func: pf.ansatz_batch(func, xx, 2_000_000, sorb, dev, torch.double)
exact = E.local_energy(x, h1e, h2e, rbm, ab, sorb, nele, noA, noB)[0]
for es in (0, 1000):
    fn = lambda: E.local_energy(x, h1e, h2e, rbm, ab, sorb, nele, noA, noB, reduce_psi=True, eps=1e-2, eps_sample=es, use_unique=True)
    r = fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    err = (r[0] - exact)
    print(f"eps_sample={es}: {ms:.2f} ms per {n} walkers ({n / ms * 1e3:.3e} E_loc/s); vs SIMPLE: mean diff {float(err.mean()):+.3e}, rms {float(err.pow(2).mean().sqrt()):.3e}")

"""The default bench step's local energies (semi-stochastic REDUCE, complex RBM) through energy.total_energy in chunks of walkers with the
look-ahead stream (front end of chunk k + 1 while the amplitudes / contraction of chunk k run) against one launch for all walkers."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from pynqs_amd import energy as E
from pynqs_amd.rbm import ComplexRBM
dev = torch.device("cuda"); torch.set_default_dtype(torch.float64)
d = B.load_fe2s2(); n = 8192
ci = d["ci_space"]
x = torch.from_numpy(np.ascontiguousarray(ci[np.arange(n) % ci.shape[0]])).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
g = torch.Generator().manual_seed(7)
m = ComplexRBM(0.02 * (torch.rand(40, 40, 2, generator=g) - 0.5), 0.02 * (torch.rand(40, 2, generator=g) - 0.5), 0.05 * (torch.rand(40, 2, generator=g) - 0.5)).to(dev)
for nb in (8192, 4096, 2048, 1024):
    for ov in (True, False):
        E.OVERLAP = ov
        fn = lambda: E.total_energy(x, nb, -1, h1, h2, m, 40, 30, 15, 15, reduce_psi=True, eps=1e-2, eps_sample=1000, dtype=torch.complex128)[0]
        for _ in range(3):
            e = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            e = fn()
        torch.cuda.synchronize()
        print(f"nbatch {nb:5d} overlap {ov}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per total_energy call; mean E {complex(e.mean()):.6f}", flush=True)

"""One GFMC step (Green's-function row + move) for Fe2S2 walkers with a real-RBM trial function: fused kernels against the
materialising path (comb + module).  usage: python tools/gfmc_time.py [n_fused] [n_module]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import gfmc, public_function as pf
from pynqs_amd.rbm import RealRBM
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/fe2s2_inputs.npz"))
dev = torch.device("cuda")
torch.set_default_dtype(torch.float64)
g = torch.Generator().manual_seed(7)
rbm = RealRBM(0.01 * (torch.rand(80, 40, generator=g) - 0.5), 0.01 * (torch.rand(80, generator=g) - 0.5), 0.1 * (torch.rand(40, generator=g) - 0.5)).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
ab = lambda x, func: pf.ansatz_batch(func, x, 2_000_000, 40, dev, torch.double)
for fused, n in ((True, int(sys.argv[1]) if len(sys.argv) > 1 else 8192), (False, int(sys.argv[2]) if len(sys.argv) > 2 else 512)):
    gfmc.FUSED_GREEN = fused
    x = torch.from_numpy(np.ascontiguousarray(d["ci_space"][:n])).to(dev)
    w = torch.ones(n, device=dev)
    def step():
        eloc, gk, comb, _, neg = gfmc.green_kernel(x, -100.0, h1, h2, rbm, ab, 40, 30, 15, 15, torch.double, None, True)
        return gfmc.sample_update(x, w, comb, gk)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        x_new, w_new, beta, acc = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{'fused' if fused else 'comb + module'}: {n} walkers, {ms:.3f} ms per step = {n / ms * 1e3:.3e} walker moves/s (accepted {acc})")

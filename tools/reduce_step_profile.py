"""One REDUCE (eps = 1e-2) local-energy step with a PyTorch module amplitude, repeated: for `rocprofv3 --kernel-trace --stats`.
usage: python tools/reduce_step_profile.py [reps] [flip|plain] [eps_sample]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import energy as E, public_function as pf
from pynqs_amd.rbm import RealRBM
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flip = len(sys.argv) > 2 and sys.argv[2] == "flip"
eps_sample = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # > 0: the semi-stochastic form of the Fe2S2 example
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/fe2s2_inputs.npz"))
dev = torch.device("cuda")
torch.set_default_dtype(torch.float64)
sorb, nele, noA, noB = 40, 30, 15, 15
g = torch.Generator().manual_seed(7)
rbm = RealRBM(0.01 * (torch.rand(2 * sorb, sorb, generator=g) - 0.5), 0.01 * (torch.rand(2 * sorb, generator=g) - 0.5),
              0.1 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
x = torch.from_numpy(np.ascontiguousarray(d["ci_space"][:8192])).to(dev)
E.FUSED_RBM = False
pf.SpinProjection.init(nele, 0)
kw = dict(use_spin_flip=True, extra_norm=torch.tensor(1.0, device=dev)) if flip else {}
fn = lambda: E.total_energy(x, 8192, 2_000_000, h1, h2, rbm, sorb, nele, noA, noB, use_unique=True, reduce_psi=True, eps=1e-2, eps_sample=eps_sample, **kw)
fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    e, _, _ = fn()
torch.cuda.synchronize()
print(f"REDUCE eps=1e-2{f' + {eps_sample} draws' if eps_sample else ''} + RBM module{' + spin flip' if flip else ''}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms per 8192 walkers, <E> = {float(e.mean()):.10f}")

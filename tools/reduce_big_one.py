"""One REDUCE configuration through energy.local_energy a few times (for rocprofv3 --kernel-trace --stats).
usage: python tools/reduce_big_one.py sorb:n_alpha:walkers:eps[:eps_sample] [onepass=1] [reps=5]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import energy as E, public_function as pf
from pynqs_amd.rbm import RealRBM

dev = torch.device("cuda")
torch.set_default_dtype(torch.float64)
f = sys.argv[1].split(":")
sorb, no, n, eps, ns = int(f[0]), int(f[1]), int(f[2]), float(f[3]), int(f[4]) if len(f) > 4 else 0
E.FUSED_ONEPASS = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
x = B.synth_walkers(n, sorb, no, no, 4321).to(dev)
h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
g = torch.Generator().manual_seed(1)
m = RealRBM(0.02 * (torch.rand(sorb, sorb, generator=g) - 0.5), 0.02 * (torch.rand(sorb, generator=g) - 0.5), 0.05 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 22, sorb, dev, torch.float64)
import time
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.time()
    e = E.local_energy(x, h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=eps, eps_sample=ns)[0]
    torch.cuda.synchronize()
    print(f"call {r}: {(time.time() - t0) * 1e3:.1f} ms (wall, with buffer allocation and growth)", flush=True)
torch.cuda.synchronize()
print("mean", float(e[torch.isfinite(e)].mean()))

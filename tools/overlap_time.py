"""total_energy, REDUCE (eps 1e-2, eps_sample 1000), 8192 Fe2S2 walkers in chunks of 2048 (the example's batch), PyTorch RealRBM on the distinct
rows: front end of chunk k + 1 on a second stream against everything on one stream (PYNQS_OVERLAP)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import energy as E
from pynqs_amd.rbm import RealRBM
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda"); torch.set_default_dtype(torch.float64)
x = torch.from_numpy(np.ascontiguousarray(d["ci_space"][:8192])).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
g = torch.Generator().manual_seed(7)
rbm = RealRBM(0.01 * (torch.rand(80, 40, generator=g) - 0.5), 0.01 * (torch.rand(80, generator=g) - 0.5), 0.1 * (torch.rand(40, generator=g) - 0.5)).to(dev)
E.FUSED_RBM = False  # the module path (what any non-RBM ansatz gets)
for ov in (True, False, True, False):
    E.OVERLAP = ov
    fn = lambda: E.total_energy(x, 2048, 2_000_000, h1, h2, rbm, 40, 30, 15, 15, reduce_psi=True, eps=1e-2, eps_sample=1000)
    torch.manual_seed(1); e0 = fn()[0]; torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"OVERLAP={ov}: {sorted(ts)[3] * 1e3:.2f} ms per 8192 walkers (4 chunks), mean E {float(e0.mean()):.6f}")

"""host-side profile of energy.total_energy on 8192 Fe2S2 walkers with the example's method (semi-stochastic REDUCE, 1000 draws) and a native
complex RBM: how far the drop-in call is from the bench step's kernels (front end + amplitudes + contraction ~0.77 ms)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynqs_amd import energy as E
from pynqs_amd.rbm import ComplexRBM
d = np.load(os.path.join(ROOT, "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda")
torch.set_default_dtype(torch.float64)
ci = d["ci_space"]
x = torch.from_numpy(np.ascontiguousarray(ci[np.arange(8192) % ci.shape[0]])).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
g = torch.Generator().manual_seed(3)
r = lambda *shape: 0.02 * (torch.rand(*shape, generator=g, dtype=torch.float64) - 0.5)  # noqa: E731
m = ComplexRBM(r(40, 40, 2), r(40, 2), r(40, 2)).to(dev)
fn = lambda: E.total_energy(x, 0, -1, h1, h2, m, 40, 30, 15, 15, reduce_psi=True, eps=1e-2, eps_sample=1000, dtype=torch.complex128)[0]  # noqa: E731
for _ in range(5):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    e = fn()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per total_energy call, finite {int(torch.isfinite(e.real).sum())}")
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    fn()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(25)

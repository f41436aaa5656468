"""Eager grad() vs GraphedGrad on the bench's amplitude module (complex128 RBM, 8192 x 40): GPU-timeline milliseconds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pynqs_amd.grad import grad, GraphedGrad
from pynqs_amd.rbm import ComplexRBM
dev = torch.device("cuda"); g = torch.Generator().manual_seed(7); sorb = 40; n = 8192
m = ComplexRBM(0.02 * (torch.rand(sorb, sorb, 2, generator=g, dtype=torch.float64) - 0.5), 0.02 * (torch.rand(sorb, 2, generator=g, dtype=torch.float64) - 0.5),
               0.05 * (torch.rand(sorb, 2, generator=g, dtype=torch.float64) - 0.5)).to(dev)
states = (torch.rand(n, sorb, device=dev, dtype=torch.float64) > 0.5).double() * 2 - 1
prob = torch.full((n,), 1.0 / n, dtype=torch.float64, device=dev)
eloc = torch.complex(torch.randn(n, device=dev, dtype=torch.float64), torch.randn(n, device=dev, dtype=torch.float64))
et = (eloc * prob).sum()
def timeit(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) / reps * 1e3
def eager():
    for p in m.parameters(): p.grad = None
    grad(m, states, prob, eloc, et, 1.0, torch.complex128, 50000)
print("eager  gpu/wall ms", timeit(eager))
g1 = [p.grad.clone() for p in m.parameters()]
gg = GraphedGrad(m, n, sorb, torch.complex128)
print("graph  gpu/wall ms", timeit(lambda: gg(states, prob, eloc, et)))
for a, p in zip(g1, m.parameters()):
    print("max rel diff", float((a - p.grad).abs().max() / a.abs().max()))

"""BASELINE-sized REDUCE through energy.total_energy (chunking by auto_nbatch, look-ahead on a second stream, table-less front end):
sorb 120, 8192 walkers (configs[2]'s shape, synthetic integrals, real RBM 120 x 120), against round 2's multi-pass path.
usage: python tools/reduce_total_energy_big.py [walkers] [eps] [eps_sample]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import energy as E
from pynqs_amd.rbm import RealRBM
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.49995
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sorb, no = 120, 30
dev = torch.device("cuda")
torch.set_default_dtype(torch.float64)
x = B.synth_walkers(n, sorb, no, no, 4321).to(dev)
h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
g = torch.Generator().manual_seed(1)
m = RealRBM(0.02 * (torch.rand(sorb, sorb, generator=g) - 0.5), 0.02 * (torch.rand(sorb, generator=g) - 0.5), 0.05 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
from pynqs_amd import public_function as pf
ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 22, sorb, dev, torch.float64)
# (the synthetic diagonal falls below eps for a few walkers: NaN there as in the reference, which total_energy refuses -- keep the others)
fin = [torch.isfinite(E.local_energy(x[b:b + 1024].contiguous(), h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=eps)[0]) for b in range(0, n, 1024)]
xs = x[torch.cat(fin)].contiguous()
res = {}
for name, onepass in (("one-launch front end (routed)", True), ("multi-pass (round 2)", False)):
    E.FUSED_ONEPASS = onepass
    def run(xs):
        return E.total_energy(xs, 0, -1, h1, h2, m, sorb, 2 * no, no, no, reduce_psi=True, eps=eps, eps_sample=ns)[0]
    nb = E.auto_nbatch(xs, h1, sorb, 2 * no, no, no, m, None, torch.double, True, ns, False, False, False, False)
    for _ in range(3):
        e = run(xs)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(3):
        e = run(xs)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    res[name] = e
    print(f"sorb {sorb}, {xs.size(0)} walkers, eps {eps}, eps_sample {ns}: {name:32s} {dt * 1e3:9.2f} ms per total_energy call = {xs.size(0) / dt:.3e} local energies/s (auto nbatch {nb})", flush=True)
if ns == 0:
    a, b = res["one-launch front end (routed)"], res["multi-pass (round 2)"]
    print(f"   max |difference| {float((a - b).abs().max()):.2e} Ha")

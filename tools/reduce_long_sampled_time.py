"""time of the semi-stochastic REDUCE front end on long rows (the flushing form; PYNQS_OP_ROW32=0: the drawn tiles enumerated a second time,
default: read back from the row's float32 copy) -- the front-end launch alone and local_energy with a real RBM, as bench.py's extras
syn56 / syn120_reduce_sample1000_local_energy.  usage: python tools/reduce_long_sampled_time.py sorb no walkers eps draws"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from pynqs_amd import energy as E, public_function as pf, reduce_front as RF
from pynqs_amd.rbm import RealRBM

sorb, no, nw, eps, ns = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda")
torch.set_default_dtype(torch.float64)
h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
x = B.synth_walkers(nw, sorb, no, no, 4321).to(dev)
g = torch.Generator().manual_seed(1)
m = RealRBM(0.02 * (torch.rand(sorb, sorb, generator=g) - 0.5), 0.02 * (torch.rand(sorb, generator=g) - 0.5), 0.05 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 22, sorb, dev, torch.float64)  # noqa: E731
fn = lambda: E.local_energy(x, h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=eps, eps_sample=ns)[0]  # noqa: E731
fn(); fn(); fn(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    e = fn()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / 3
fe = next(iter(E._FRONTS.values()))
plan = E.CX.plan_for(h1, h2, sorb, dev)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(5):
    fe.run(x, plan.buf, eps, 100 + i, None)
b.record(); b.synchronize()
print(f"sorb {sorb} x {nw} walkers, eps {eps}, {ns} draws (ROW32={os.environ.get('PYNQS_OP_ROW32', '1')}): local_energy {el * 1e3:.3f} ms, front end alone {a.elapsed_time(b) / 5:.3f} ms; "
      f"row_f32 {'yes' if fe.row_f32 is not None else 'no'} (form {fe.row_f32_form}), tile scratch {'yes' if fe.tile_scratch is not None else 'no'}, table {'yes' if fe.dedup else 'no'}, "
      f"finite {int(torch.isfinite(e).sum())}, mean {float(e[torch.isfinite(e)].mean()):.6f}")

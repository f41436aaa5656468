"""pynqs_rbm_forward_children against pynqs_rbm_forward on the distinct x' of a REDUCE front end at sizes whose factor table exceeds the LDS
(a wave per row).  usage: python tools/rbm_children_big.py [sorb n_alpha walkers eps H kind]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import C_extension as cx, energy as E
a = sys.argv[1:]
sorb, no, n, eps = (int(a[0]), int(a[1]), int(a[2]), float(a[3])) if len(a) >= 4 else (120, 30, 4096, 0.49995)
H = int(a[4]) if len(a) > 4 else sorb
kind = a[5] if len(a) > 5 else "real"
dev = torch.device("cuda")
x = B.synth_walkers(n, sorb, no, no, 4321).to(dev)
h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
fe, nu = E.reduce_front(x, h1, h2, sorb, 2 * no, no, no, eps, 0, want_pm1=False)
g = torch.Generator().manual_seed(1)
r = lambda *s: (0.04 * (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)).to(dev)
W, hb, vb = (r(H, sorb, 2), r(H, 2), r(sorb, 2)) if kind == "complex" else (r(H, sorb), r(H), r(sorb))
uniq = fe.uniq_onv[:nu].contiguous()

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        out = fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps, out

t0, want = timeit(lambda: cx.rbm_forward(uniq, W, hb, vb, sorb, kind))
t1, got = timeit(lambda: cx.rbm_forward_children(uniq, fe.uniq_parent, x, W, hb, vb, sorb, kind))
err = float(((got - want).abs() / want.abs().clamp_min(1e-300)).max())
print(f"sorb {sorb}, {H} hidden units ({kind}), {nu} rows of {n} walkers: from scratch {t0:.3f} ms, from the parents {t1:.3f} ms (table build included); max relative difference {err:.2e}")

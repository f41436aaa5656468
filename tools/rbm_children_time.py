"""pynqs_rbm_children_prepare + pynqs_rbm_forward_children on the distinct x' of 8192 Fe2S2 walkers (semi-stochastic REDUCE front end,
complex RBM with 40 hidden units) against pynqs_rbm_forward on the same rows: time per call, agreement."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx, reduce_front as RF
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda"); n = 8192
ci = d["ci_space"]
x = torch.from_numpy(np.ascontiguousarray(ci[np.arange(n) % ci.shape[0]])).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
plan = cx.plan_for(h1, h2, 40, dev).buf
fe = RF.ReduceFrontEnd(n, 40, 30, 15, 15, 1000, torch.float64, dev, 246, 1900000, want_pm1=False)
fe.run(x, plan, 1e-2, 3, None)
nu = int(fe.counters[0])
g = torch.Generator().manual_seed(7)
W = (0.02 * (torch.rand(40, 40, 2, generator=g, dtype=torch.float64) - 0.5)).to(dev)
hb = (0.02 * (torch.rand(40, 2, generator=g, dtype=torch.float64) - 0.5)).to(dev)
vb = (0.05 * (torch.rand(40, 2, generator=g, dtype=torch.float64) - 0.5)).to(dev)
out = torch.zeros(fe.cap_unique, dtype=torch.complex128, device=dev)
def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
t_c = timed(lambda: cx.rbm_forward_children(fe.uniq_onv, fe.uniq_parent, x, W, hb, vb, 40, "complex", count=fe.counters, out=out))
t_f = timed(lambda: cx.rbm_forward(fe.uniq_onv[:nu], W, hb, vb, 40, "complex"))
want = cx.rbm_forward(fe.uniq_onv[:nu].contiguous(), W, hb, vb, 40, "complex")
rel = float(((out[:nu] - want).abs() / want.abs()).max())
print(f"{nu} distinct rows: from parents {t_c:.1f} us, from scratch {t_f:.1f} us, max relative difference {rel:.2e}")

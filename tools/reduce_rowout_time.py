"""time of the semi-stochastic front end on 8192 Fe2S2 walkers (PYNQS_OP_DEBUG / PYNQS_OP_ROWLDS ablations)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynqs_amd import C_extension as cx, reduce_front as RF
d = np.load(os.path.join(ROOT, "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda"); n = int(os.environ.get("NW", "8192")); N = int(os.environ.get("NS", "1000")); eps = float(os.environ.get("EPS", "1e-2"))
ci = d["ci_space"]
x = torch.from_numpy(np.ascontiguousarray(ci[np.arange(n) % ci.shape[0]])).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
plan = cx.plan_for(h1, h2, 40, dev).buf
fe = RF.ReduceFrontEnd(n, 40, 30, 15, 15, N, torch.float64, dev, 246, 1900000, want_pm1=False)
for _ in range(5):
    fe.run(x, plan, eps, 3, None)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(50):
    fe.run(x, plan, eps, 3 + i, None)
b.record(); b.synchronize()
print(f"DEBUG={os.environ.get('PYNQS_OP_DEBUG', '0')} N={N}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per {n} walkers, counters {fe.counters.tolist()}")

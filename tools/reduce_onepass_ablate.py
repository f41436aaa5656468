"""PYNQS_OP_DEBUG ablations of the one-launch REDUCE front end (timing only; 8192 Fe2S2 walkers)."""
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
for N, capu in ((0, 300000), (1000, 1900000)):
    fe = RF.ReduceFrontEnd(n, 40, 30, 15, 15, N, torch.float64, dev, 246, capu, want_pm1=os.environ.get("NO_PM1") != "1")
    for _ in range(3):
        fe.run(x, plan, 1e-2, 3, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fe.run(x, plan, 1e-2, 3, None)
    b.record(); b.synchronize()
    print(f"PYNQS_OP_DEBUG={os.environ.get('PYNQS_OP_DEBUG', '0')} eps_sample {N}: {a.elapsed_time(b) / 20 * 1e3:.1f} us  counters {fe.counters.tolist()}")

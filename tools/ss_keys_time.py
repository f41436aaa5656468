"""Key-major SAMPLE_SPACE kernel alone (native entry), 8192 Fe2S2 walkers against the benchmark's table (18496 keys, complex psi) or K random keys."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx, _native as N
d = np.load("tests/golden/fe2s2_inputs.npz")
dev = torch.device("cuda")
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x = torch.from_numpy(d["ci_space"][:n].copy()).to(dev)
if K == 0:
    keys = torch.from_numpy(d["ci_space"].copy()).to(dev)
else:
    g = torch.Generator().manual_seed(5)
    occ = torch.zeros((K, 40), dtype=torch.uint8)
    for s in (0, 1):
        occ.scatter_(1, 2 * torch.rand(K, 20, generator=g).argsort(1)[:, :15] + s, 1)
    keys = torch.unique(torch.cat([cx.tensor_to_onv(occ.to(dev), 40), x]), dim=0)
wf = torch.rand(keys.size(0), dtype=torch.float64, device=dev) + 0.1
plan = cx.plan_for(h1e, h2e, 40, dev)
eloc = torch.empty(n, dtype=torch.float64, device=dev); psi0 = torch.empty(n, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
call = lambda: N.check(N.lib().pynqs_eloc_sample_space_keys(x.data_ptr(), n, 40, 30, 15, 15, plan.data_ptr(), keys.data_ptr(), keys.size(0), wf.data_ptr(), 0, 0,
                                                            eloc.data_ptr(), psi0.data_ptr(), st), "keys")
for _ in range(3):
    call()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    call()
b.record(); torch.cuda.synchronize()
print(f"n={n} keys={keys.size(0)}: {a.elapsed_time(b) / 20:.3f} ms per call, <E> = {float(eloc.mean()):.8f}")

"""get_comb_tensor + get_hij_torch(x, comb_x) as the reference's REDUCE / SAMPLE_SPACE code calls them (eloc.py:243-252): the generic pair
kernel against the recognised-list path (fused plan kernel, Hmat only)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx
d = np.load("tests/golden/fe2s2_inputs.npz")
dev = torch.device("cuda")
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
x = torch.from_numpy(d["ci_space"][:8192].copy()).to(dev)
for reuse in (True, False):
    cx.REUSE_COMB = reuse
    def step():
        comb, _ = cx.get_comb_tensor(x, 40, 30, 15, 15)
        return cx.get_hij_torch(x, comb, h1e, h2e, 40, 30)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        h = step()
    torch.cuda.synchronize()
    print(f"get_comb_tensor + get_hij_torch, 8192 walkers, recognised list = {reuse}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")

"""get_hij_torch in 2-D (CI-matrix) mode: cost of the diagonal pairs, which one lane evaluates serially (465 ordered terms for Fe2S2)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx
d = np.load("tests/golden/fe2s2_inputs.npz")
dev = torch.device("cuda")
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
n = 4096
x = torch.from_numpy(d["ci_space"][:n].copy()).to(dev)
y = torch.from_numpy(d["ci_space"][n:2 * n].copy()).to(dev)
def t(a, b):
    cx.get_hij_torch(a, b, h1e, h2e, 40, 30); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        cx.get_hij_torch(a, b, h1e, h2e, 40, 30)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3
print(f"{n} x {n} pairs, bra == ket ({n} diagonal pairs): {t(x, x):.3f} ms;  disjoint bra / ket (no diagonal pair): {t(x, y):.3f} ms")

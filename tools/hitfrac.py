"""Which fraction of the connected determinants x' of Fe2S2 walkers lies in the sample table (ci_space)?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx, public_function as pf
d = np.load("tests/golden/fe2s2_inputs.npz")
dev = torch.device("cuda")
ci = torch.from_numpy(d["ci_space"].copy()).to(dev)
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
lut = pf.WavefunctionLUT(ci, torch.ones(ci.size(0), dtype=torch.float64, device=dev), 40, device=dev)
for a, b in ((0, 256), (4000, 4256), (18000, 18256)):
    comb, hm = cx.get_comb_hij_fused(ci[a:b].contiguous(), h1e, h2e, 40, 30, 15, 15)
    idx, mask = cx.hash_lookup(lut.hashtable, comb.reshape(-1, 8))
    print(f"walkers {a}:{b}: hit fraction {float(mask.float().mean()):.4f}, nonzero |H|>1e-12 fraction {float((hm.abs() > 1e-12).float().mean()):.4f}")

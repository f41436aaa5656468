"""Secondary kernels of the path, timed with events on the Fe2S2 problem (run on the GPU box).
Prints one line per kernel: time, bytes moved (algorithmic), GB/s."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import C_extension as cx
from pynqs_amd.public_function import WavefunctionLUT

d = np.load("tests/golden/fe2s2_inputs.npz")
dev = torch.device("cuda")
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
ci = torch.from_numpy(d["ci_space"]).to(dev)
x = ci[:2048].contiguous()
torch.set_default_dtype(torch.float64)

def t(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps

comb, hm = cx.get_comb_hij_fused(x, h1, h2, 40, 30, 15, 15)
flat = comb.reshape(-1, 8)
n = flat.size(0)
rows = []
ms = t(lambda: cx.onv_to_tensor(flat, 40)); rows.append(("onv_to_tensor f64", ms, n * (8 + 40 * 8)))
torch.set_default_dtype(torch.float32)
ms = t(lambda: cx.onv_to_tensor(flat, 40)); rows.append(("onv_to_tensor f32", ms, n * (8 + 40 * 4)))
torch.set_default_dtype(torch.float64)
occ = (cx.onv_to_tensor(flat[:4_000_000], 40) > 0).to(torch.uint8)
ms = t(lambda: cx.tensor_to_onv(occ, 40)); rows.append(("tensor_to_onv", ms, occ.size(0) * (40 + 8)))
ms = t(lambda: cx.get_comb_tensor(x, 40, 30, 15, 15)); rows.append(("get_comb_tensor", ms, n * 8))
ms = t(lambda: cx.get_hij_torch(x, comb, h1, h2, 40, 30)); rows.append(("get_hij_torch 3-D", ms, n * 16))
k = ci[:4096].contiguous()
ms = t(lambda: cx.get_hij_torch(k, k, h1, h2, 40, 30)); rows.append(("get_hij_torch 2-D 4096^2", ms, 4096 * 4096 * 16))
lut = WavefunctionLUT(ci, torch.rand(ci.size(0), dtype=torch.float64, device=dev), 40, device=dev)
ms = t(lambda: cx.wavefunction_lut(lut.bra_key, flat, 40)); rows.append(("wavefunction_lut 18496 keys", ms, n * 17))
ms = t(lambda: cx.spin_flip_rand(flat[:4_000_000], 40, 30, 15, 15, 7)); rows.append(("spin_flip_rand (+onv_to_tensor)", ms, 4_000_000 * (16 + 320)))
for name, ms, b in rows:
    print(f"{name:34s} {ms:9.4f} ms   {b/1e6:10.1f} MB   {b/ms/1e6:8.1f} GB/s   ({n} rows)" )

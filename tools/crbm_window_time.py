"""The complex-parameter RBM kernel (pynqs_eloc_crbm) windowed: synthetic sorb-120 walkers (30 alpha, 30 beta; ncomb 1.19e6) with 240 complex hidden units, and Fe2S2 with forced windows
against the resident rows.  usage: python tools/crbm_window_time.py [walkers at sorb 120]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import C_extension as cx

def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

dev = torch.device("cuda")
g = torch.Generator().manual_seed(7)
def table(sorb, H):
    r = lambda *s: (0.02 * (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)).to(dev)
    return cx.CRBMTable(r(H, sorb, 2), r(H, 2), r(sorb, 2))

d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fe2s2_inputs.npz"))
x = torch.from_numpy(d["ci_space"][:8192].copy()).to(dev)
h1e, h2e = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
tab = table(40, 80)
for w in ("0", "40", "16"):
    if w == "0": os.environ.pop("PYNQS_CRBM_WINDOW", None)
    else: os.environ["PYNQS_CRBM_WINDOW"] = w
    ms = timed(lambda: cx.eloc_crbm(x, h1e, h2e, tab, 40, 30, 15, 15), 5)
    print(f"Fe2S2, 8192 walkers, 80 complex hidden units, window {w if w != '0' else 'none (resident rows)'}: {ms:.3f} ms", flush=True)
os.environ.pop("PYNQS_CRBM_WINDOW", None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sorb, no, H = 120, 30, 240
xs = B.synth_walkers(n, sorb, no, no, 4321).to(dev)
h1, h2 = B.synth_integrals(sorb)
h1, h2 = h1.to(dev), h2.to(dev)
tab = table(sorb, H)
ms = timed(lambda: cx.eloc_crbm(xs, h1, h2, tab, sorb, 2 * no, no, no), 2)
ncomb = 1 + 2 * no * (sorb // 2 - no) + (no * (sorb // 2 - no)) ** 2 + 2 * (no * (no - 1) // 2) * ((sorb // 2 - no) * (sorb // 2 - no - 1) // 2)
print(f"sorb 120, {n} walkers, {H} complex hidden units (windowed): {ms:.1f} ms per launch = {n / ms * 1e3:.3e} E_loc/s; "
      f"{n * ncomb * H * 11 / ms / 1e9:.2f} T f64 instruction-lanes/s (11 per column and hidden unit)", flush=True)

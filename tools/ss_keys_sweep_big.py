"""The two fused SAMPLE_SPACE kernels against the size of the table at sorb 120 / 184 (half filling, synthetic integrals, bench.py's
generators): where does walking the table stop paying?  usage: python tools/ss_keys_sweep_big.py [sorb] [walkers]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from pynqs_amd import energy, public_function as pf

sorb = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
no = {56: 7, 120: 30, 184: 46}[sorb]
dev = torch.device("cuda")
h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
x_cpu = B.synth_walkers(n, sorb, no, no, 4321)
x = x_cpu.to(dev)
ncomb = pf.get_Num_SinglesDoubles(sorb, no, no) + 1
print(f"sorb {sorb}, {n} walkers, ncomb {ncomb}", flush=True)
for logk in (14, 16, 18, 20, 22):
    K = 1 << logk
    keys = torch.unique(torch.cat([x_cpu, B.synth_connected(x_cpu, sorb, K - n, 99)]), dim=0).to(dev)
    wf = torch.rand(keys.size(0), dtype=torch.float64, device=dev) + 0.1
    lut = pf.WavefunctionLUT(keys, wf, sorb, device=dev)
    f = lambda: energy.local_energy(x, h1, h2, None, None, sorb, 2 * no, no, no, WF_LUT=lut, use_sample_space=True)
    out = {}
    for mode in (True, False):
        energy.SS_KEYS = mode
        e = f()[0]; torch.cuda.synchronize()
        reps = 5 if mode else 2
        t0 = time.perf_counter()
        for _ in range(reps):
            e = f()[0]
        torch.cuda.synchronize()
        out[mode] = ((time.perf_counter() - t0) / reps * 1e3, e)
    err = float((out[True][1] - out[False][1]).abs().max())
    print(f"keys 2^{logk} ({keys.size(0)}; {keys.size(0) / ncomb:.3f} x ncomb): key-major {out[True][0]:.3f} ms, column-major {out[False][0]:.3f} ms   "
          f"(max |dE| between them {err:.1e})", flush=True)
    del lut, keys, wf

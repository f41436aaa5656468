"""Semi-stochastic selection (energy.reduce_compact_sampled) on a synthetic sorb-120 system: time per call."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pynqs_amd import energy as E
sorb, no, n = int(sys.argv[1]) if len(sys.argv) > 1 else 120, None, int(sys.argv[2]) if len(sys.argv) > 2 else 256
no = {56: 7, 120: 30, 184: 46}[sorb]
dev = torch.device("cuda")
h1, h2 = bench.synth_integrals(sorb)
h1, h2 = h1.to(dev), h2.to(dev)
x = bench.synth_walkers(n, sorb, no, no, 4321).to(dev)
for eps_sample in (1000, 100):
    f = lambda: E.reduce_compact_sampled(x, h1, h2, sorb, 2 * no, no, no, 0.495, eps_sample, seed=5)
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        kept, drawn = f()
    torch.cuda.synchronize()
    print(f"sorb {sorb}, {n} walkers, eps_sample {eps_sample}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms; kept {kept[0].numel()}, drawn records {drawn[0].numel()}, "
          f"sum of weights/walker {float(drawn[3].abs().sum() / n):.4f}")

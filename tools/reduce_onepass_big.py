"""The one-launch REDUCE front end at the BASELINE shapes beyond Fe2S2: synthetic sorb 120 (30 alpha, 30 beta; ncomb 1.19e6) and sorb 184
(46, 46; ncomb 6.6e6), deterministic and semi-stochastic (N = 1000), against the multi-pass path.  usage: python tools/reduce_onepass_big.py [walkers120] [walkers184]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import energy as E, C_extension as cx, reduce_front as RF

dev = torch.device("cuda")
E.FRONT_ROUTE = False  # (this tool times the one-launch front end itself, whatever energy.py would route)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

EPS = float(os.environ.get("EPS", "0.495"))
for sorb, no, n, eps in ((120, 30, int(sys.argv[1]) if len(sys.argv) > 1 else 256, EPS), (184, 46, int(sys.argv[2]) if len(sys.argv) > 2 else 64, EPS)):
    x = B.synth_walkers(n, sorb, no, no, 4321).to(dev)
    h1, h2 = B.synth_integrals(sorb)
    h1, h2 = h1.to(dev), h2.to(dev)
    plan = cx.plan_for(h1, h2, sorb, dev).buf
    ncomb = int(cx.get_Num_SinglesDoubles(sorb, no, no)) + 1
    for N in (0, 1000):
        ok = RF.supported(n, sorb, 2 * no, no, no, N)
        if not ok:
            print(f"sorb {sorb}, {n} walkers, eps_sample {N}: the fused form does not fit (LDS): multi-pass path")
            continue
        fe, nu = E.reduce_front(x, h1, h2, sorb, 2 * no, no, no, eps, N, seed=3, want_pm1=False)
        cnt = fe.counters_host()
        t = timeit(lambda: fe.run(x, plan, eps, 3, None))
        print(f"sorb {sorb}, {n} walkers x {ncomb} columns, eps {eps}, eps_sample {N}: front end {t:.3f} ms = {n / t * 1e3:.3e} walkers/s = {n * ncomb / t * 1e3 / 1e9:.1f} G columns/s; "
              f"{nu} distinct x', row cache {'yes' if fe.row_cache is not None else 'no'}, chunks per walker {fe.nchunks}", flush=True)
    t2 = timeit(lambda: E.reduce_compact(x, h1, h2, sorb, 2 * no, no, no, eps), 3)
    print(f"   multi-pass reduce_compact (count + emit + glue): {t2:.3f} ms", flush=True)

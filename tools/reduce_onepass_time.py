"""Timing of the one-launch REDUCE front end against the multi-pass path of round 2 (8192 Fe2S2 walkers)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import energy as E, C_extension as cx

d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ci = d["ci_space"]
x = torch.from_numpy(np.ascontiguousarray(ci[np.arange(n) % ci.shape[0]])).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
sorb, nele, noA, noB = 40, 30, 15, 15
plan = cx.plan_for(h1, h2, sorb, dev).buf

def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps

for N in (0, 1000):
    fe, nu = E.reduce_front(x, h1, h2, sorb, nele, noA, noB, 1e-2, N, seed=3)
    cnt = fe.counters_host()
    rec = fe.records()
    print(f"eps_sample {N}: distinct {nu}, records {rec[0].numel()}, max kept doubles/seg {cnt[2]}, caps d={fe.cap_doubles} u={fe.cap_unique} slots={fe.dedup_slots}")
    t = timeit(lambda: fe.run(x, plan, 1e-2, 3, None))
    amp = torch.rand(fe.cap_unique, dtype=torch.float64, device=dev) + 0.5
    tc = timeit(lambda: fe.contract(amp))
    print(f"   front end {t*1e3:.1f} us   contract {tc*1e3:.1f} us")
    if N == 0:
        t2 = timeit(lambda: E.reduce_compact(x, h1, h2, sorb, nele, noA, noB, 1e-2))
        print(f"   round-2 reduce_compact (count + emit + glue, host sync) {t2*1e3:.1f} us")
    else:
        torch.manual_seed(0)
        t2 = timeit(lambda: E.reduce_compact_sampled(x, h1, h2, sorb, nele, noA, noB, 1e-2, N, seed=3), 5)
        print(f"   round-2 reduce_compact_sampled {t2*1e3:.1f} us")

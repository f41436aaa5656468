"""REDUCE local energies through energy.local_energy with a real RBM on the distinct x': the one-launch front end against the multi-pass
path of round 2 (energy.FUSED_ONEPASS = False), on synthetic integrals of any size or on Fe2S2 -- the measurement behind energy's
routing rule.
usage: python tools/reduce_big_paths.py CASE [CASE ...]     CASE = sorb:n_alpha:walkers:eps[:eps_sample]   (sorb 40 with n_alpha 15 = Fe2S2)
environment: ROUTE=1 also times the default routing (energy.FUSED_ONEPASS = True with the rule on)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import energy as E, public_function as pf
from pynqs_amd.rbm import RealRBM

dev = torch.device("cuda")
torch.set_default_dtype(torch.float64)


def timeit(fn, reps=3):
    fn(); fn(); fn(); torch.cuda.synchronize()   # (the first call sizes the buffers, the second may switch the de-duplication off)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps, out


for case in sys.argv[1:]:
    f = case.split(":")
    sorb, no, n, eps, ns = int(f[0]), int(f[1]), int(f[2]), float(f[3]), int(f[4]) if len(f) > 4 else 0
    if sorb == 40 and no == 15:
        d = B.load_fe2s2()
        x = torch.from_numpy(np.ascontiguousarray(d["ci_space"][:n])).to(dev)
        h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
    else:
        x = B.synth_walkers(n, sorb, no, no, 4321).to(dev)
        h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
    g = torch.Generator().manual_seed(1)
    m = RealRBM(0.02 * (torch.rand(sorb, sorb, generator=g) - 0.5), 0.02 * (torch.rand(sorb, generator=g) - 0.5), 0.05 * (torch.rand(sorb, generator=g) - 0.5)).to(dev)
    ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 22, sorb, dev, torch.float64)
    res = {}
    modes = [("one-launch front end", True, False), ("multi-pass (round 2)", False, False)]
    if os.environ.get("ROUTE"):
        modes.append(("routed (default)", True, True))
    for name, onepass, route in modes:
        E.FUSED_ONEPASS = onepass
        E.FRONT_ROUTE = route   # (False: the one-launch front end whatever the row length)
        E._FRONTS.clear()
        E._FRONT_DENSE.clear()
        E._FRONT_NODEDUP.clear()
        fn = lambda: E.local_energy(x, h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=eps, eps_sample=ns)[0]
        t, e = timeit(fn)
        res[name] = e
        where = " (took the multi-pass path)" if route and not E._front_ok(x, h1, sorb, 2 * no, no, no, ns) else ""
        if route and any(v is not None for v in E._FRONT_NODEDUP.values()):
            where += " (without the de-duplication table)"
        print(f"sorb {sorb}, {n} walkers, eps {eps}, eps_sample {ns}: {name:22s} {t:9.3f} ms per call{where}", flush=True)
    a, b = res[modes[0][0]], res[modes[1][0]]
    both = torch.isfinite(a) & torch.isfinite(b)
    print(f"   max |difference| {float((a - b)[both].abs().max()):.2e} Ha on {int(both.sum())} walkers; not finite: {int((~torch.isfinite(a)).sum())} / {int((~torch.isfinite(b)).sum())}"
          f" (same walkers: {bool((torch.isfinite(a) == torch.isfinite(b)).all())})", flush=True)

"""Randomised check of the one-launch REDUCE front end's table-less / flushing forms against the multi-pass kernels (reduce_compact):
random (sorb, noA, noB, walkers, eps, dtype), deterministic and with draws; records must be identical, every record must own a row with
its determinant.  usage: python tools/fuzz_flush.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from pynqs_amd import C_extension as cx, energy as E, reduce_front as RF

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda")
t0, cases, flushed, sampled_cases = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    sorb = int(rng.choice([4, 6, 8, 12, 16, 20, 24, 30, 36, 40, 56, 66, 70, 80, 130]))
    K = sorb // 2
    cap = 4 if sorb > 66 else (8 if sorb > 40 else K)
    noA, noB = int(rng.integers(0, min(K, cap) + 1)), int(rng.integers(0, min(K, cap) + 1))
    if noA + noB == 0:
        continue
    n = int(rng.choice([1, 3, 17, 64, 300, 1200, 5000]))
    ncomb = int(cx.get_Num_SinglesDoubles(sorb, noA, noB)) + 1
    if n * ncomb > 3e8:
        n = max(1, int(3e8 // ncomb))
    dt = torch.float64 if rng.random() < 0.7 else torch.float32
    x = B.synth_walkers(n, sorb, noA, noB, int(rng.integers(1 << 30))).to(dev)
    h1, h2 = (t.to(dev).to(dt) for t in B.synth_integrals(sorb, int(rng.integers(1 << 30))))
    eps = float(rng.choice([1e-12, 0.05, 0.2, 0.4, 0.47, 0.49, 0.499]))
    ns = int(rng.choice([0, 0, 7, 100])) if ncomb > 10 else 0
    row, col2, onv2, hh, counts = E.reduce_compact(x, h1, h2, sorb, noA + noB, noA, noB, eps, sort=True)
    if row.numel() > 3e7:
        continue   # (torch's advanced indexing of [1e8, 16] uint8 tensors returned garbage rows on this stack: keep the harness below that)
    cap_d = int(counts.max()) + int(rng.integers(0, 9))
    dedup = bool(rng.random() < 0.3)
    if not RF.supported(n, sorb, noA + noB, noA, noB, ns) or cap_d > RF.list_capacity(n, sorb, noA + noB, noA, noB, ns, dt, without_table=not dedup):
        continue
    fe = RF.ReduceFrontEnd(n, sorb, noA + noB, noA, noB, ns, dt, dev, cap_d, int(counts.sum()) + n * ns + 64, want_pm1=False, dedup=dedup)
    fe.run(x, cx.plan_for(h1, h2, sorb, dev).buf, eps, seed=int(rng.integers(1 << 40)))
    nu, flags, _ = fe.counters_host()
    assert flags == 0, (sorb, noA, noB, n, eps, ns, flags)
    w, col, h, link, onv, drawn = fe.records()
    kw, kc, kh, ko = w[~drawn], col[~drawn], h[~drawn], onv[~drawn]
    k1 = torch.argsort((kw << 32) | kc.long(), stable=True)
    ok = kw.numel() == row.numel() and torch.equal(kw[k1], row) and torch.equal(kc[k1], col2) and torch.equal(kh[k1], hh) and torch.equal(ko[k1], onv2)
    assert ok, ("kept records differ", sorb, noA, noB, n, eps, ns, str(dt), dedup)
    rows = fe.rows_of(link)
    assert torch.equal(fe.uniq_onv[rows], onv), ("rows", sorb, noA, noB, n, eps, ns)
    if not dedup:
        assert nu == w.numel() and torch.unique(rows).numel() == nu
    if ns:
        sampled_cases += 1
        tot = torch.zeros(n, dtype=torch.float64, device=dev)
        S = fe.row_sum[:n]
        dw = w[drawn]
        cnt = (h[drawn].double().abs() * ns / S[dw]).round()
        tot.index_add_(0, dw, cnt)
        has = S > 0
        assert bool((tot[has] == ns).all()) and bool((tot[~has] == 0).all()), ("draw counts", sorb, noA, noB, n, eps, ns)
    flushed += int(cap_d + fe.fixed > 2048)
    cases += 1
    del fe
print(f"fuzz ok: {cases} systems in {time.time() - t0:.0f} s ({flushed} with more than one flush per segment, {sampled_cases} with draws)")

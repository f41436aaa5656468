"""Randomised differential test of the round-4 semi-stochastic REDUCE kernel (kernels_reduce_rowout.hip: rows of up to 8192 columns) against the
CPU oracle: random (sorb, noA, noB, walkers, eps, draws, integral dtype, with / without de-duplication table); eps = 0 (nothing kept: every
column can be drawn), eps so large that little is left to draw, more than 1024 draws (the parked-columns path), rows without any sub-eps
width.  Checked per system: kept set / values / kets exactly the oracle's |H| >= eps, S to 1e-12 relative, every drawn record a sub-eps column
of non-zero element with a whole hit count, counts adding up to N per walker, weights (c / N) sign(H) S, no column twice, ascending columns,
links leading to the records' determinants, and the same seed giving the same records.
usage: python tools/fuzz_draws.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from pynqs_amd import C_extension as cx, reduce_front as RF, _native as N_

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda")
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731


def integrals(sorb, scale):
    h1 = rng.random((sorb, sorb)) - 0.5
    h1 = (h1 + h1.T).reshape(-1)
    pair = sorb * (sorb - 1) // 2
    h2 = rng.random(pair * (pair + 1) // 2) - 0.5
    if rng.random() < 0.15:   # sparse integrals: many exact zeros among the widths
        h2 *= rng.random(h2.shape) < 0.3
    return h1 * scale, h2 * scale


def walkers(n, sorb, noA, noB):
    occ = np.zeros((n, sorb), dtype=np.uint8)
    for i in range(n):
        occ[i, 2 * rng.permutation(sorb // 2)[:noA]] = 1
        occ[i, 2 * rng.permutation(sorb // 2)[:noB] + 1] = 1
    return O.pm01_to_onv(occ, sorb)


t0, cases, forms = time.time(), 0, 0
while time.time() - t0 < budget:
    sorb = int(rng.choice([4, 6, 8, 10, 12, 14, 16, 20, 24, 30, 36, 40, 66, 70, 130]))
    K = sorb // 2
    cap = 4 if sorb > 40 else (7 if sorb > 24 else K)
    noA = int(rng.integers(0, min(K, cap) + 1)); noB = int(rng.integers(0, min(K, cap) + 1))
    ncomb = N_.lib().pynqs_num_sd(sorb, noA, noB) + 1
    if noA + noB == 0 or ncomb < 2 or ncomb > 8192:
        continue
    n = int(rng.integers(1, 40))
    ns = int(rng.choice([1, 7, 64, 200, 1000, 1025, 2500]))
    scale = float(rng.choice([1.0, 1e-3, 30.0]))
    eps = float(rng.choice([0.0, 0.05, 0.3, 0.45, 0.6])) * scale
    f32 = bool(rng.random() < 0.25)
    dedup = bool(rng.random() < 0.8)
    h1, h2 = integrals(sorb, scale)
    if f32:
        h1, h2 = h1.astype(np.float32), h2.astype(np.float32)
    xh = walkers(n, sorb, noA, noB)
    x, h1g, h2g = G(xh), G(h1), G(h2)
    dt = torch.float32 if f32 else torch.float64
    co, ho = O.comb_hij_fused(xh, h1, h2, sorb, noA + noB, noA, noB)
    ho = torch.from_numpy(ho)
    keep = (ho.abs() >= eps) if eps > 0 else torch.zeros_like(ho, dtype=torch.bool)
    kept_max = int(keep.sum(1).max())
    fe = RF.ReduceFrontEnd(n, sorb, noA + noB, noA, noB, ns, dt, dev, kept_max + 4, n * (kept_max + ns) + 64, want_pm1=False, dedup=dedup)
    plan = cx.plan_for(h1g, h2g, sorb, dev).buf
    seed = int(rng.integers(0, 2**62))
    fe.run(x, plan, eps, seed, None)
    nu, flags, _ = fe.counters_host()
    assert flags == 0, (sorb, noA, noB, n, eps, ns, flags)
    forms += fe.row_f32_form == 1
    walker, col, w, link, onv, drawn = fe.records()
    wk, cl, ww, dr, ov = walker.cpu(), col.cpu().long(), w.cpu(), drawn.cpu(), onv.cpu()
    tag = f"sorb {sorb} {noA}a{noB}b n {n} eps {eps} N {ns} f32 {f32} dedup {dedup} seed {seed}"
    got = torch.zeros_like(keep)
    got[wk[~dr], cl[~dr]] = True
    assert torch.equal(got, keep), "kept set: " + tag
    assert torch.equal(ww[~dr], ho[wk[~dr], cl[~dr]]), "kept values: " + tag
    if int(cl.max()) >= ho.shape[1]:
        bad = cl >= ho.shape[1]
        print("column out of range:", tag, "ncomb", ho.shape[1], "cols", cl[bad][:8].tolist(), "drawn", dr[bad][:8].tolist(), "w", ww[bad][:8].tolist(),
              "row tail", fe.row_f32.view(n, -1)[int(wk[bad][0])][-20:].tolist())
        raise SystemExit(1)
    kets = torch.from_numpy(co).reshape(n, ho.shape[1], -1)[wk, cl]
    assert torch.equal(ov, kets), "kets: " + tag
    rows = fe.rows_of(link).cpu()
    assert torch.equal(fe.uniq_onv.cpu()[rows], kets), "links: " + tag
    sub = torch.where(keep, torch.zeros_like(ho), ho.abs()).double()
    S = sub.sum(1)
    rs = fe.row_sum[:n].cpu()
    assert bool(((rs - S).abs() <= (1e-12 if not f32 else 1e-6) * S.abs() + 1e-300).all()), "S: " + tag
    dw, dc, dh = wk[dr], cl[dr], ww[dr].double()
    assert not bool(keep[dw, dc].any()) and bool((ho[dw, dc] != 0).all()), "drawn columns: " + tag
    flat = dw * ho.shape[1] + dc
    # (ascending columns: the round-4 kernel's order; the other forms -- taken when the kept records outgrow 1024 slots: the flushing form with
    # or without the row's float32 copy, row_f32_form 2 / 0 -- emit tile by tile)
    if not (flat.unique().numel() == flat.numel() and (fe.row_f32_form != 1 or bool((flat[1:] > flat[:-1]).all()))):
        badi = int((flat[1:] <= flat[:-1]).nonzero()[0])
        print("drawn order:", tag, "ncomb", ho.shape[1], "unique", flat.unique().numel(), "of", flat.numel(), "at", badi, "walkers", dw[badi - 2: badi + 4].tolist(),
              "cols", dc[badi - 2: badi + 4].tolist(), "w", dh[badi - 2: badi + 4].tolist())
        wb = int(dw[badi + 1])
        mine = dc[dw == wb]
        print("walker", wb, "kept", int(keep[wb].sum()), "seg_count", int(fe.seg_count[wb]), "fixed", fe.fixed, "cap_d", fe.cap_doubles, "drawn records", mine.numel(), "first 40 cols", mine[:40].tolist())
        print("srec_col slots of the walker (first 60)", fe.srec_col.view(n, -1)[wb][:60].tolist())
        raise SystemExit(1)
    hits = dh * ns / (torch.sign(ho[dw, dc]).double() * rs[dw])
    assert bool(((hits - hits.round()).abs() < (1e-6 if not f32 else 2e-2)).all()) and bool((hits.round() >= 1).all()), "hit counts: " + tag
    tot = torch.zeros(n, dtype=torch.float64).index_add_(0, dw, hits.round())
    # a float32 width can vanish where the float64 element does not (|h| < 1e-45 x ...): such a row still draws N times from what is left
    drawable = (sub.float() > 0).any(1) if True else None
    assert bool((tot[drawable] == ns).all()) and bool((tot[~drawable] == 0).all()), "N draws per walker: " + tag
    rec = (fe.srec_col.clone(), fe.srec_w.clone(), fe.rec_col.clone())
    fe.run(x, plan, eps, seed, None)
    torch.cuda.synchronize()
    used = rec[0] >= 0
    assert torch.equal(rec[0], fe.srec_col) and torch.equal(rec[1][used], fe.srec_w[used]) and torch.equal(rec[2], fe.rec_col), "same seed, same records: " + tag
    cases += 1
print(f"fuzz_draws ok: {cases} systems in {time.time() - t0:.0f} s ({forms} through the round-4 kernel for short rows)")

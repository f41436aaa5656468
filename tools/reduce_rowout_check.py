"""LDS-row form of the semi-stochastic REDUCE front end (round 4) on 8192 Fe2S2 walkers: structure of its records against the CPU oracle's
rows on the first walkers (kept records exact, drawn records sub-eps with whole hit counts adding up to N, weights (c / N) sign(H) S),
the law of the draws (z-scores of the per-column hit counts over many seeds), and its time.  PYNQS_OP_ROWLDS=0 runs the row-cache form."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pynqs_amd import C_extension as cx, reduce_front as RF
from oracle import oracle as O

d = np.load(os.path.join(ROOT, "tests", "golden", "fe2s2_inputs.npz"))
dev = torch.device("cuda"); n = int(os.environ.get("NW", "8192")); N = int(os.environ.get("NS", "1000")); eps = float(os.environ.get("EPS", "1e-2"))
ci = d["ci_space"]
xh = np.ascontiguousarray(ci[np.arange(n) % ci.shape[0]])
x = torch.from_numpy(xh).to(dev)
h1, h2 = torch.from_numpy(d["h1e"]).to(dev), torch.from_numpy(d["h2e"]).to(dev)
plan = cx.plan_for(h1, h2, 40, dev).buf
fe = RF.ReduceFrontEnd(n, 40, 30, 15, 15, N, torch.float64, dev, 246, 1900000, want_pm1=False)
print("row_f32", fe.row_f32 is not None, "row_cache", fe.row_cache is not None)
fe.run(x, plan, eps, 3, None)
torch.cuda.synchronize()
print("counters", fe.counters.tolist())
m = 8
walker, col, w, link, onv, drawn = fe.records()
sel = walker < m
wk, cl, ww, dr = walker[sel].cpu(), col[sel].cpu().long(), w[sel].cpu(), drawn[sel].cpu()
co, ho = O.comb_hij_fused(xh[:m], d["h1e"], d["h2e"], 40, 30, 15, 15)
ho = torch.from_numpy(ho)
keep = ho.abs() >= eps
got = torch.zeros_like(keep); got[wk[~dr], cl[~dr]] = True
print("kept set exact:", bool(torch.equal(got, keep)), " kept values exact:", bool(torch.equal(ww[~dr], ho[wk[~dr], cl[~dr]])))
S = torch.where(keep, torch.zeros_like(ho), ho.abs()).sum(1)
print("row_sum max rel diff:", float(((fe.row_sum[:m].cpu() - S) / S).abs().max()))
hits = ww[dr].abs() * N / fe.row_sum[:m].cpu()[wk[dr]]
print("hits whole:", float((hits - hits.round()).abs().max()), " none kept:", not bool(keep[wk[dr], cl[dr]].any()),
      " signs:", bool(torch.equal(torch.sign(ww[dr]), torch.sign(ho[wk[dr], cl[dr]]))))
tot = torch.zeros(m, dtype=torch.float64).index_add_(0, wk[dr], hits.round())
print("hits per walker:", tot.tolist())
# ascending columns within a walker's drawn records
ok = True
for i in range(m):
    c = cl[dr][wk[dr] == i]
    ok = ok and bool((c[1:] > c[:-1]).all())
print("drawn columns ascending:", ok)
kets = torch.from_numpy(co).reshape(m, ho.shape[1], -1)[wk, cl]
rows = fe.rows_of(link[sel]).cpu()
print("links lead to the determinants:", bool(torch.equal(fe.uniq_onv.cpu()[rows], kets)))
# the law: accumulate hit counts of walker 0..m-1 over many seeds, z-scores against N_total p_j
reps = int(os.environ.get("REPS", "200"))
acc = torch.zeros(m, ho.shape[1], dtype=torch.float64)
for s in range(reps):
    fe.run(x, plan, eps, 1000 + s, None)
    sc = fe.srec_col[: m * N].view(m, N).cpu().long()
    sw = fe.srec_w[: m * N].view(m, N).cpu()
    rs = fe.row_sum[:m].cpu()
    for i in range(m):
        v = sc[i] >= 0
        acc[i].index_add_(0, sc[i][v], (sw[i][v].abs() * N / rs[i]).round())
p = torch.where(keep, torch.zeros_like(ho), ho.abs()) / S[:, None]
exp = p * N * reps
z = (acc - exp) / torch.sqrt(exp * (1 - p) + 1e-30)
zz = z[p > 0]
print(f"law over {reps} seeds: hits total {acc.sum(1).tolist()[:3]}..., z mean {float(zz.mean()):.3f} std {float(zz.std()):.3f} max |z| {float(zz.abs().max()):.2f} ({zz.numel()} columns)")
# time
for _ in range(3):
    fe.run(x, plan, eps, 3, None)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for i in range(50):
    fe.run(x, plan, eps, 3 + i, None)
b.record(); b.synchronize()
print(f"front end: {a.elapsed_time(b) / 50 * 1e3:.1f} us per {n} walkers, counters {fe.counters.tolist()}")

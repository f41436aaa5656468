"""REDUCE on long rows (more than energy.FRONT_LONG_ROW columns; the reference's _reduce_psi, vmc/energy/eloc.py:205-324):
* the flushing LIST form of the one-launch front end (the list is emptied as it fills) against the multi-pass kernels, record by record;
* the table-less mode (no de-duplication when nearly all x' are distinct) against the generic tensor path;
* the routing of the semi-stochastic form: the front end while a segment's records fit its list, the multi-pass path afterwards."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(sorb, no, n):
    import bench as B
    from pynqs_amd.rbm import RealRBM

    dev = torch.device("cuda")
    x = B.synth_walkers(n, sorb, no, no, 99).to(dev)
    h1, h2 = (t.to(dev) for t in B.synth_integrals(sorb))
    g = torch.Generator().manual_seed(1)
    r = lambda *s: 0.05 * (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)  # noqa: E731
    return x, h1, h2, RealRBM(r(sorb // 2, sorb), r(sorb // 2), r(sorb)).to(dev)


@pytest.mark.parametrize("sorb,no,n,eps", [(80, 20, 300, 0.3), (80, 20, 5000, 0.49), (136, 4, 4100, 0.47)])
def test_flushing_list_form_writes_the_records_of_the_multi_pass_kernels(sorb, no, n, eps):
    """(col, <x|H|x'>, x') of every kept column, walker by walker in ascending columns (column 0 and the unpaired doubles of tile 0 first),
    bit for bit; the distinct list holds every x' once; rows cut into chunks (300 walkers) and whole rows (5000)."""
    from pynqs_amd import energy as E, reduce_front as RF

    x, h1, h2, _ = _case(sorb, no, n)
    assert E.get_Num_SinglesDoubles(sorb, no, no) + 1 > E.FRONT_LONG_ROW
    assert RF.list_capacity(n, sorb, 2 * no, no, no, 0) == (1 << 30) - 1
    fe, nu = E.reduce_front(x, h1, h2, sorb, 2 * no, no, no, eps, want_pm1=False)
    assert fe.cap_doubles + fe.fixed > 2048   # (more than a list holds: several flushes per segment)
    w, col, h, link, onv, _ = fe.records()
    row, col2, onv2, h2_, counts = E.reduce_compact(x, h1, h2, sorb, 2 * no, no, no, eps, sort=True)
    assert w.numel() == row.numel() and int(counts.sum()) == w.numel()
    k1 = torch.argsort((w << 32) | col.long(), stable=True)
    assert torch.equal(w[k1], row) and torch.equal(col[k1], col2) and torch.equal(h[k1], h2_) and torch.equal(onv[k1], onv2)
    # inside a segment: ascending columns after the entries of tile 0
    c = fe.rec_col.view(fe.nseg, fe.stride)[0, : fe.fixed + int(fe.seg_count[0])].long()
    c = c[c >= 0]
    assert int((c[1:] < c[:-1]).sum()) <= 6   # (at most the few unpaired doubles in front are out of order)
    # links: every record points at a row holding its determinant; rows are distinct
    rows = fe.rows_of(link)
    assert torch.equal(fe.uniq_onv[rows], onv)
    assert torch.unique(fe.uniq_onv[:nu], dim=0).size(0) == nu


@pytest.mark.parametrize("system,eps,dedup", [("fe2s2", 1e-12, False), ("fe2s2", 1e-3, False), ("sorb56", 0.47, True), ("sorb56", 0.47, False), ("sorb56", 0.2, False)])
def test_short_rows_in_the_flushing_form_and_without_the_table(system, eps, dedup, fe2s2):
    """Short rows take the flushing form when at most a tenth of the columns is kept (sorb 56, eps 0.47: 6 %) and whenever there is no
    de-duplication table (ReduceFrontEnd(dedup=False): every record its own row): records as the multi-pass kernels write them."""
    import numpy as np
    from pynqs_amd import C_extension as cx, energy as E, reduce_front as RF

    dev = torch.device("cuda")
    if system == "fe2s2":
        sorb, no, n = 40, 15, 4096
        x = torch.from_numpy(np.ascontiguousarray(fe2s2["ci_space"][:n])).to(dev)
        h1, h2 = torch.from_numpy(fe2s2["h1e"]).to(dev), torch.from_numpy(fe2s2["h2e"]).to(dev)
    else:
        sorb, no, n = 56, 7, 4100
        x, h1, h2, _ = _case(sorb, no, n)
    row, col2, onv2, h2_, counts = E.reduce_compact(x, h1, h2, sorb, 2 * no, no, no, eps, sort=True)
    cap_d = int(counts.max()) + 8
    assert cap_d <= RF.list_capacity(n, sorb, 2 * no, no, no, 0, without_table=not dedup)
    assert cap_d + RF.geometry(n, sorb, 2 * no, no, no, 0)[1] > 1024      # (not the plain LIST form)
    fe = RF.ReduceFrontEnd(n, sorb, 2 * no, no, no, 0, torch.float64, dev, cap_d, int(counts.sum()) + 64, want_pm1=False, dedup=dedup)
    fe.run(x, cx.plan_for(h1, h2, sorb, dev).buf, eps)
    nu, flags, _ = fe.counters_host()
    assert flags == 0
    w, col, h, link, onv, _ = fe.records()
    assert w.numel() == row.numel() == fe.count_records()
    k1 = torch.argsort((w << 32) | col.long(), stable=True)
    assert torch.equal(w[k1], row) and torch.equal(col[k1], col2) and torch.equal(h[k1], h2_) and torch.equal(onv[k1], onv2)
    rows = fe.rows_of(link)
    assert torch.equal(fe.uniq_onv[rows], onv)
    if dedup:
        assert torch.unique(fe.uniq_onv[:nu], dim=0).size(0) == nu   # (a row's parent: whichever walker's record inserted it first)
    else:
        assert nu == w.numel() and torch.unique(rows).numel() == nu and bool((link >= (1 << 30)).all())
        assert torch.equal(fe.uniq_parent[rows].long(), w)
    # the contraction on these records: sum_k h_k A(x'_k) / A(x) with A = a function of the determinant's bytes
    a = (fe.uniq_onv[:nu].double() * torch.arange(1, fe.uniq_onv.size(1) + 1, device=dev).double()).sum(1).cos() + 2.0
    eloc, ax = fe.contract(a)
    a_rec = (onv2.double() * torch.arange(1, onv2.size(1) + 1, device=dev).double()).sum(1).cos() + 2.0
    first = torch.zeros(n, dtype=torch.float64, device=dev)
    first[row[col2 == 0]] = a_rec[col2 == 0]
    want = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, row, h2_ * a_rec) / first
    ok = first != 0
    assert bool(ok.any()) and float((eloc - want)[ok].abs().max()) < 1e-9 * float(want[ok].abs().max())


def test_long_rows_drop_the_table_when_everything_is_distinct(monkeypatch):
    from pynqs_amd import energy as E, public_function as pf

    sorb, no, n = 80, 20, 320
    x, h1, h2, m = _case(sorb, no, n)
    dev = x.device
    ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 20, sorb, dev, torch.float64)  # noqa: E731
    for name, val in (("_FRONT_DENSE", set()), ("_FRONTS", {}), ("_FRONT_NODEDUP", {})):
        monkeypatch.setattr(E, name, val)

    def energies(eps, xs=x):
        return E.local_energy(xs, h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=eps)[0]

    def check(e, eps):
        monkeypatch.setattr(E, "FUSED", False)
        want = energies(eps, x[:6].contiguous())
        monkeypatch.setattr(E, "FUSED", True)
        ok = torch.isfinite(want)
        assert bool((torch.isfinite(e[:6]) == ok).all()) and bool(ok.any())
        assert float((e[:6] - want)[ok].abs().max()) < 1e-8  # Ha

    e1 = energies(0.3)             # first call: with the table; nearly every x' turns out distinct
    check(e1, 0.3)
    (key, caps), = E._FRONT_NODEDUP.items()
    assert caps is not None and not E._FRONTS
    e2 = energies(0.3)             # table-less from now on
    fe = next(iter(E._FRONTS.values()))
    assert fe.table is None and not fe.dedup
    both = torch.isfinite(e1)
    assert bool((torch.isfinite(e2) == both).all()) and float((e2 - e1)[both].abs().max()) < 1e-10
    e3 = energies(0.3)
    assert torch.equal(torch.nan_to_num(e3), torch.nan_to_num(e2))
    # total_energy (look-ahead tickets, two workspaces) on the walkers whose diagonal survives eps
    fin = both.nonzero().flatten()[:256]
    for _ in range(2):
        et = E.total_energy(x[fin].contiguous(), 128, -1, h1, h2, m, sorb, 2 * no, no, no, reduce_psi=True, eps=0.3)[0]
        assert float((et - e1[fin]).abs().max()) < 1e-9
    # both look-ahead workspaces of the 128-walker chunks ended up table-less
    slots = [f for k, f in E._FRONTS.items() if k[1] == 128]
    assert len(slots) == 2 and not any(f.dedup for f in slots)
    # Fe2S2-like duplication keeps the table: walkers repeated four times
    monkeypatch.setattr(E, "_FRONT_NODEDUP", {})
    monkeypatch.setattr(E, "_FRONTS", {})
    xr = x[:64].repeat(4, 1).contiguous()
    er = energies(0.3, xr)
    (key, caps), = E._FRONT_NODEDUP.items()
    assert caps is None and next(iter(E._FRONTS.values())).dedup
    ok = torch.isfinite(er[:64])
    assert float((er[:64] - e1[:64])[ok].abs().max()) < 1e-10


def test_semi_stochastic_long_rows_leave_the_front_end_when_the_list_overflows(monkeypatch):
    from pynqs_amd import energy as E, public_function as pf, reduce_front as RF

    sorb, no, n, ns = 80, 20, 128, 200
    x, h1, h2, m = _case(sorb, no, n)
    dev = x.device
    ab = lambda xx, func: pf.ansatz_batch(func, xx, 1 << 20, sorb, dev, torch.float64)  # noqa: E731
    for name, val in (("_FRONT_DENSE", set()), ("_FRONTS", {}), ("_FRONT_NODEDUP", {})):
        monkeypatch.setattr(E, name, val)
    assert not E._front_ok(x, h1, sorb, 2 * no, no, no, ns)   # (few walkers with draws on long rows: the multi-pass path)
    monkeypatch.setattr(E, "FRONT_SAMPLED_MIN_WALKERS", 1)
    # (the routing exists for rows the flushing form cannot serve -- its LDS does not fit, PYNQS_OP_FLUSH=0: emulated by a capacity
    # query that knows the plain LIST form only)
    monkeypatch.setattr(RF, "list_capacity", lambda *a, **k: 2048 - RF.geometry(n, sorb, 2 * no, no, no, ns)[1])
    limit = RF.list_capacity(n, sorb, 2 * no, no, no, ns)
    assert 0 < limit < 2048 and E._front_ok(x, h1, sorb, 2 * no, no, no, ns)
    calls = {"front": 0, "multi": 0}
    run, compact = RF.ReduceFrontEnd.run, E.reduce_compact_sampled
    monkeypatch.setattr(RF.ReduceFrontEnd, "run", lambda self, *a, **k: (calls.__setitem__("front", calls["front"] + 1), run(self, *a, **k))[1])
    monkeypatch.setattr(E, "reduce_compact_sampled", lambda *a, **k: (calls.__setitem__("multi", calls["multi"] + 1), compact(*a, **k))[1])

    def energies(eps):
        torch.manual_seed(5)
        return E.local_energy(x, h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=eps, eps_sample=ns)[0]

    e1 = energies(0.4999)          # a few dozen kept columns per row: the LIST form
    assert calls["front"] >= 1 and calls["multi"] == 0 and not E._FRONT_DENSE
    nf = calls["front"]
    e2 = energies(0.47)            # thousands: the list overflows, the call and the following ones take the multi-pass path
    assert calls["multi"] == 1 and calls["front"] == nf + 1 and len(E._FRONT_DENSE) == 1
    assert not E._front_ok(x, h1, sorb, 2 * no, no, no, ns)
    e3 = energies(0.47)
    assert calls["multi"] == 2 and calls["front"] == nf + 1
    assert E._front_ok(x, h1, sorb, 2 * no, no, no, 0)   # (the deterministic form has its own key)
    # same walkers without a diagonal above eps (NaN there, as in the reference), finite elsewhere
    ed = E.local_energy(x, h1, h2, m, ab, sorb, 2 * no, no, no, reduce_psi=True, eps=0.47)[0]
    assert bool((torch.isfinite(e2) == torch.isfinite(ed)).all()) and bool((torch.isfinite(e3) == torch.isfinite(ed)).all())
    assert int(torch.isfinite(e1).sum()) > n // 2


@pytest.mark.parametrize("sorb,no,n,eps,ns", [(80, 20, 48, 0.47, 300), (136, 4, 40, 0.45, 100), (56, 7, 64, 0.47, 100), (56, 7, 64, 0.47, 1000),
                                              (80, 20, 24, 0.47, 3000)])   # (3000 draws: more than fit behind the draw slots, the re-enumerating form)
def test_semi_stochastic_flushing_form(sorb, no, n, eps, ns):
    """Long rows with draws and more kept columns than the list holds: the kept list is flushed during the enumeration, the draws follow.
    Kept records = the multi-pass kernels', bit for bit; every drawn record is a sub-eps column with weight (c / N) sign(H) S, S = the
    row's sum of sub-eps |H| (eloc.py:257-298), and the counts of a row add up to N."""
    from pynqs_amd import C_extension as cx, energy as E, reduce_front as RF

    x, h1, h2, _ = _case(sorb, no, n)
    long_row = E.get_Num_SinglesDoubles(sorb, no, no) + 1 > E.FRONT_LONG_ROW   # (sorb 56: a short row, sparse enough for the flushing form)
    assert RF.list_capacity(n, sorb, 2 * no, no, no, ns) == ((1 << 30) - 1 if long_row else 30976 // 10)
    fe, nu = E.reduce_front(x, h1, h2, sorb, 2 * no, no, no, eps, ns, seed=17, want_pm1=False)
    assert fe.cap_doubles + fe.fixed > 2048 and (fe.tile_scratch is not None) == long_row
    # (end of round 4: the draws read the drawn tiles back from the row's float32 copy instead of enumerating them again)
    assert (fe.row_f32 is not None) == (fe.row_f32_form == 2) and fe.row_cache is None and (fe.row_f32_form == 2) == (ns <= 1000)
    assert fe.cap_doubles <= RF.list_capacity(n, sorb, 2 * no, no, no, ns)
    w, col, h, link, onv, drawn = fe.records()
    row, col2, onv2, h2_, counts = E.reduce_compact(x, h1, h2, sorb, 2 * no, no, no, eps, sort=True)
    kw, kc, kh, ko = w[~drawn], col[~drawn], h[~drawn], onv[~drawn]
    k1 = torch.argsort((kw << 32) | kc.long(), stable=True)
    assert torch.equal(kw[k1], row) and torch.equal(kc[k1], col2) and torch.equal(kh[k1], h2_) and torch.equal(ko[k1], onv2)
    comb, hm = cx.get_comb_hij_fused(x, h1, h2, sorb, 2 * no, no, no)
    sub = torch.where(hm.abs() < eps, hm.abs(), torch.zeros_like(hm))
    S = sub.sum(1)
    assert float((fe.row_sum[:n] - S).abs().max()) <= 1e-12 * float(S.max())
    dw, dc, dh, do = w[drawn], col[drawn].long(), h[drawn], onv[drawn]
    hd = hm[dw, dc]
    assert bool((hd.abs() < eps).all()) and bool((hd != 0).all())
    cnt = dh * ns / (torch.sign(hd) * fe.row_sum[dw])           # the multiplicity c of the record
    assert float((cnt - cnt.round()).abs().max()) < 1e-9 and bool((cnt.round() >= 1).all())
    tot = torch.zeros(n, dtype=torch.float64, device=x.device).index_add_(0, dw, cnt.round())
    assert bool((tot == ns).all())
    assert torch.equal(do, comb[dw, dc])
    assert torch.equal(fe.uniq_onv[fe.rows_of(link)], onv)
    # a column is drawn about as often as its share of S says (loose: chi-square-free check on the largest share)
    share = (sub / S.unsqueeze(1)).max(1)
    big = torch.zeros(n, dtype=torch.float64, device=x.device)
    hit = dc == share.indices[dw]
    big.index_add_(0, dw[hit], cnt.round()[hit])
    assert float((big / ns - share.values).abs().max()) < 0.05


def test_tile_sums_in_global_memory_draw_the_same_records(monkeypatch):
    """Semi-stochastic LIST form on long rows: with io->tile_scratch the per-tile sums and draw counts live in global memory (two workgroups
    per CU at sorb 120 instead of one); same kept records, same draws, same weights, bit for bit."""
    from pynqs_amd import C_extension as cx, reduce_front as RF

    for sorb, no, n, eps, ns in ((80, 20, 200, 0.4995, 300), (136, 4, 100, 0.499, 64)):
        x, h1, h2, _ = _case(sorb, no, n)
        plan = cx.plan_for(h1, h2, sorb, x.device).buf
        out = []
        for min_row in (65536, 1 << 40):
            monkeypatch.setattr(RF, "TILE_SCRATCH_MIN_ROW", min_row)
            fe = RF.ReduceFrontEnd(n, sorb, 2 * no, no, no, ns, torch.float64, x.device, 600, 3000 * n, want_pm1=False)
            assert (fe.tile_scratch is not None) == (min_row == 65536)
            fe.run(x, plan, eps, seed=11)
            nu, flags, _ = fe.counters_host()
            assert flags == 0
            w, col, h, link, onv, drawn = fe.records()
            assert int(drawn.sum()) > n * ns // 2
            out.append((nu, w, col, h, onv, drawn, fe.row_sum.clone()))
        a, b = out
        assert a[0] == b[0] and all(torch.equal(u, v) for u, v in zip(a[1:], b[1:]))


def test_short_rows_stay_on_the_front_end(fe2s2):
    from pynqs_amd import energy as E

    x = torch.from_numpy(fe2s2["ci_space"][:64].copy()).cuda()
    h1 = torch.from_numpy(fe2s2["h1e"]).cuda()
    assert E._long_row_cap(64, h1, 40, 30, 15, 15, 0) is None and E._front_ok(x, h1, 40, 30, 15, 15, 0) and E._front_ok(x, h1, 40, 30, 15, 15, 100)


@pytest.mark.parametrize("sorb,no,n,eps,ns,dedup", [
    (80, 20, 300, 0.3, 0, True),         # flushing LIST form, rows cut into chunks
    (80, 20, 5000, 0.49, 0, True),       # flushing LIST form, whole rows
    (136, 4, 4100, 0.47, 0, False),      # table-less (three words)
    (120, 30, 64, 0.4995, 0, True),      # BASELINE configs[2]'s shape: 30 alpha 30 beta, two words, 1.19e6 columns
    (184, 46, 8, 0.4999, 0, True),       # configs[4]'s shape: 46 alpha 46 beta, three words, 6.6e6 columns
    (80, 20, 48, 0.47, 300, True),       # semi-stochastic flushing form, tile sums in global memory
    (136, 4, 40, 0.45, 100, True),
    (120, 30, 16, 0.4995, 200, True),
    (184, 46, 4, 0.4999, 100, True),
    (56, 7, 64, 0.47, 1000, True),       # short rows, sparse: flushing with draws
])
def test_long_row_forms_against_the_oracle(sorb, no, n, eps, ns, dedup):
    """The direct leg for the forms above (the other tests compare them with the multi-pass kernels of this library, which share
    plan_tiles.h / plan_dev.h with them): on the first walkers of the batch the kept records are exactly |<x|H|x'>| >= eps of the CPU
    oracle's row (oracle/pynqs_oracle.c, pinned to the reference bit for bit) -- set, values and kets --, the drawn records are sub-eps
    columns of that row with whole hit counts adding up to N and weights (c / N) sign(H) S (vmc/energy/eloc.py:257-298), and every
    link leads to the record's determinant."""
    import numpy as np
    from oracle import oracle as O
    from pynqs_amd import C_extension as cx, energy as E, reduce_front as RF

    x, h1, h2, _ = _case(sorb, no, n)
    dev = x.device
    m = min(n, 4 if sorb < 150 else 2)
    if dedup:
        E._FRONTS.clear()
        fe, nu = E.reduce_front(x, h1, h2, sorb, 2 * no, no, no, eps, ns, seed=23, want_pm1=False)
    else:
        _, _, _, _, counts = E.reduce_compact(x, h1, h2, sorb, 2 * no, no, no, eps, sort=True)
        fe = RF.ReduceFrontEnd(n, sorb, 2 * no, no, no, ns, torch.float64, dev, int(counts.max()) + 8, int(counts.sum()) + 64, want_pm1=False, dedup=False)
        fe.run(x, cx.plan_for(h1, h2, sorb, dev).buf, eps, 23)
        assert fe.counters_host()[1] == 0
    walker, col, w, link, onv, drawn = fe.records()
    sel = walker < m
    rows = fe.rows_of(link[sel])
    assert torch.equal(fe.uniq_onv[rows], onv[sel])
    walker, col, w, onv, drawn = walker[sel].cpu(), col[sel].cpu().long(), w[sel].cpu(), onv[sel].cpu(), drawn[sel].cpu()
    co, ho = O.comb_hij_fused(x[:m].cpu().numpy(), h1.cpu().numpy(), h2.cpu().numpy(), sorb, 2 * no, no, no)
    ho = torch.from_numpy(ho)
    keep = ho.abs() >= eps
    got = torch.zeros_like(keep)
    got[walker[~drawn], col[~drawn]] = True
    assert torch.equal(got, keep), "kept set differs from the oracle's |H| >= eps"
    assert torch.equal(w[~drawn], ho[walker[~drawn], col[~drawn]]), "kept values differ from the oracle's"
    kets = torch.from_numpy(co).reshape(m, ho.shape[1], -1)[walker, col]
    assert torch.equal(onv, kets), "a record's determinant is not the oracle's x'"
    if ns:
        S = torch.where(keep, torch.zeros_like(ho), ho.abs()).sum(1)
        assert float(((fe.row_sum[:m].cpu() - S) / S).abs().max()) < 1e-12
        assert not bool(keep[walker[drawn], col[drawn]].any()) and bool((ho[walker[drawn], col[drawn]] != 0).all())
        hits = w[drawn] * ns / (torch.sign(ho[walker[drawn], col[drawn]]) * S[walker[drawn]])
        assert float((hits - hits.round()).abs().max()) < 1e-6 and bool((hits.round() >= 1).all())
        tot = torch.zeros(m, dtype=torch.float64).index_add_(0, walker[drawn], hits.round())
        assert bool((tot == ns).all())
        flat = walker[drawn] * ho.shape[1] + col[drawn]
        assert flat.unique().numel() == flat.numel()
    else:
        assert not bool(drawn.any())

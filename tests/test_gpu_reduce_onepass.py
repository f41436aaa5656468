"""The one-launch REDUCE front end (pynqs_reduce_onepass / pynqs_reduce_contract, pynqs_amd.reduce_front) against
  * the materialised matrix of the drop-in kernel (itself bit-identical to the reference, tests/test_gpu_parity_core.py): the kept
    records are exactly |<x|H|x'>| >= eps (vmc/energy/eloc.py:297-298), values and kets bit for bit, 1-3 word determinants, chunked rows;
  * the definition of the semi-stochastic selection (eloc.py:257-296): distinct sub-eps columns, hits add up to N, weights
    (hits / N) sign(H) S, multinomial z-scores (fixed seeds: deterministic test);
  * `Func`'s de-duplication (vmc/energy/flip.py:44-50): the distinct list is the set of the records' determinants, every record's
    link leads to its determinant, the +-1 rows are onv_to_tensor of the list;
  * the contraction against a plain torch evaluation of sum_k w_k psi(x'_k) / psi(x) (tolerance 1e-10 relative: a rounded sum)."""
import numpy as np
import pytest
import torch

from conftest import golden, rand_occ, synth_integrals

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _setup(sorb, noA, noB, n, seed, dtype=torch.float64):
    from pynqs_amd import C_extension as cx

    h1, h2 = synth_integrals(sorb)
    h1e, h2e = _dev(h1).to(dtype), _dev(h2).to(dtype)
    x = cx.tensor_to_onv(_dev(rand_occ(n, sorb, noA, noB, seed=seed)), sorb)
    comb, hm = cx.get_comb_hij_fused(x, h1e, h2e, sorb, noA + noB, noA, noB)
    return x, h1e, h2e, comb, hm


def _front(x, h1e, h2e, sorb, noA, noB, eps, N=0, lut=None, seed=11):
    from pynqs_amd import energy

    energy._FRONTS.clear()
    return energy.reduce_front(x, h1e, h2e, sorb, noA + noB, noA, noB, eps, N, lut, seed=seed)


def _check_distinct(fe, nu, walker, onv, link, sorb):
    """the distinct list and the links (records that do not point into a wave-function table)"""
    from pynqs_amd import C_extension as cx

    uniq = fe.uniq_onv[:nu]
    own = link >= 0
    rows = fe.rows_of(link[own])
    assert int(rows.min()) >= 0 and int(rows.max()) < nu
    assert torch.equal(uniq[rows], onv[own])                                    # every record finds its determinant
    want = torch.unique(onv[own], dim=0)
    assert want.size(0) == nu and torch.equal(torch.unique(uniq, dim=0), want)  # each determinant once
    assert torch.equal(fe.uniq_pm1[:nu], cx.onv_to_tensor(uniq, sorb).to(fe.uniq_pm1.dtype))
    return rows, own


@pytest.mark.parametrize("sorb,noA,noB,n,eps", [(40, 15, 15, 64, 1e-2), (12, 3, 2, 24, 0.2), (16, 4, 4, 24, 0.0), (66, 3, 3, 16, 0.3),
                                                 (130, 2, 2, 16, 0.25), (120, 30, 30, 3, 0.495), (184, 46, 46, 2, 0.498),
                                                 (12, 3, 3, 5000, 0.3)])
def test_kept_records_are_exact(sorb, noA, noB, n, eps, fe2s2):
    if sorb == 40:
        from pynqs_amd import C_extension as cx

        x = _dev(fe2s2["ci_space"][:n])
        h1e, h2e = _dev(fe2s2["h1e"]), _dev(fe2s2["h2e"])
        comb, hm = cx.get_comb_hij_fused(x, h1e, h2e, sorb, noA + noB, noA, noB)
    else:
        x, h1e, h2e, comb, hm = _setup(sorb, noA, noB, n, seed=sorb)
    fe, nu = _front(x, h1e, h2e, sorb, noA, noB, eps)
    walker, col, w, link, onv, drawn = fe.records()
    assert not bool(drawn.any())
    keep = hm.abs() >= eps
    order = torch.argsort(walker * hm.size(1) + col.long())
    r2, c2 = torch.where(keep)
    assert torch.equal(walker[order], r2) and torch.equal(col[order].long(), c2)
    assert torch.equal(w[order], hm[keep]) and torch.equal(onv[order], comb[keep])
    assert int(fe.seg_count[: fe.nseg].sum()) + int((fe.rec_col.view(fe.nseg, fe.stride)[:, : fe.fixed] >= 0).sum()) == int(keep.sum())
    _check_distinct(fe, nu, walker, onv, link, sorb)
    # reproducible: same positions, same values from run to run (the distinct list's ORDER may differ: it follows the atomics)
    a = [t.clone() for t in (fe.rec_col, fe.rec_w, fe.seg_count)]
    fe.run(x, __import__("pynqs_amd").C_extension.plan_for(h1e, h2e, sorb, x.device).buf, eps, 0, None)
    w2 = fe.records()
    assert torch.equal(w2[0], walker) and torch.equal(w2[1], col) and torch.equal(w2[2], w) and torch.equal(w2[4], onv)
    assert torch.equal(a[2], fe.seg_count)
    # contraction against torch, real and complex amplitudes on the distinct rows
    g = torch.Generator(device="cpu").manual_seed(5)
    for cplx in (False, True):
        nu2 = fe.counters_host()[0]
        amp = (torch.rand(nu2, generator=g, dtype=torch.float64) + 0.5).cuda()
        if cplx:
            amp = amp * torch.exp(1j * torch.rand(nu2, generator=g, dtype=torch.float64).cuda())
        e, px = fe.contract(amp)
        wk2, col2, ww, lk, _, _ = fe.records()
        rows = fe.rows_of(lk)
        num = torch.zeros(n, dtype=amp.dtype, device="cuda").index_add_(0, wk2, ww.to(amp.dtype) * amp[rows])
        first = col2 == 0
        p0 = torch.zeros(n, dtype=amp.dtype, device="cuda")
        p0[wk2[first]] = amp[rows[first]]
        if eps > 0 and bool((hm[:, 0].abs() < eps).any()):
            continue  # (psi(x) = 0 rows: inf / nan on both sides)
        assert torch.equal(px, p0)
        torch.testing.assert_close(e, num / p0, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("sorb,noA,noB,eps,n", [(12, 3, 2, 0.2, 24), (16, 4, 4, 0.35, 24), (66, 3, 3, 0.3, 24), (130, 2, 2, 0.25, 24),
                                                 (12, 3, 3, 0.0, 24), (40, 15, 15, 1e-2, 48)])
def test_drawn_records_follow_the_definition(sorb, noA, noB, eps, n, fe2s2):
    N = 4000 if sorb != 40 else 1000
    if sorb == 40:
        from pynqs_amd import C_extension as cx

        x = _dev(fe2s2["ci_space"][100:100 + n])
        h1e, h2e = _dev(fe2s2["h1e"]), _dev(fe2s2["h2e"])
        comb, hm = cx.get_comb_hij_fused(x, h1e, h2e, sorb, noA + noB, noA, noB)
    else:
        x, h1e, h2e, comb, hm = _setup(sorb, noA, noB, n, seed=sorb)
    fe, nu = _front(x, h1e, h2e, sorb, noA, noB, eps, N, seed=77)
    walker, col, w, link, onv, drawn = fe.records()
    keep = hm.abs() >= eps if eps > 0 else torch.zeros_like(hm, dtype=torch.bool)
    # kept part
    k = ~drawn
    got = torch.zeros_like(keep)
    got[walker[k], col[k].long()] = True
    assert torch.equal(got, keep) and torch.equal(w[k], hm[walker[k], col[k].long()]) and torch.equal(onv[k], comb[walker[k], col[k].long()])
    # drawn part: distinct sub-eps columns, right kets, right signs, hits add up to N per row
    s_row, s_col, s_w = walker[drawn], col[drawn].long(), w[drawn]
    assert not keep[s_row, s_col].any()
    flat = s_row * hm.size(1) + s_col
    assert flat.unique().numel() == flat.numel()
    assert torch.equal(onv[drawn], comb[s_row, s_col])
    S = torch.where(keep, torch.zeros_like(hm), hm.abs()).sum(1)
    torch.testing.assert_close(fe.row_sum[:n], S, rtol=1e-13, atol=0)
    hits = s_w.abs() * N / S[s_row]
    assert torch.allclose(hits, hits.round(), atol=1e-6) and bool((hits.round() >= 1).all())
    assert torch.equal(torch.sign(s_w), torch.sign(hm[s_row, s_col]))
    tot = torch.zeros(n, dtype=torch.float64, device=x.device).index_add_(0, s_row, hits.round())
    assert torch.equal(tot, torch.full_like(tot, float(N)))
    # distribution: hits ~ Binomial(N, p_m) per column; pooled z-scores stay within 6 sigma
    p = torch.where(keep, torch.zeros_like(hm), hm.abs()) / S[:, None]
    emp = torch.zeros_like(hm)
    emp[s_row, s_col] = hits.round()
    z = (emp - N * p) / torch.sqrt(N * p * (1 - p) + 1e-12)
    sel = p > 5.0 / N
    if bool(sel.any()):
        assert float(z[sel].abs().max()) < 6.0
        assert abs(float(z[sel].mean())) < 6.0 / np.sqrt(float(sel.sum()))
    nb = hm.size(1) // 64
    if nb:
        eb = emp[:, : nb * 64].reshape(n, nb, 64).sum(-1)
        pb = p[:, : nb * 64].reshape(n, nb, 64).sum(-1)
        zb = (eb - N * pb) / torch.sqrt(N * pb * (1 - pb) + 1e-12)
        selb = pb > 5.0 / N
        if bool(selb.any()):
            assert float(zb[selb].abs().max()) < 6.0
    _check_distinct(fe, nu, walker, onv, link, sorb)
    # same seed -> same records; another seed -> other draws
    rec1 = [t.clone() for t in (fe.srec_col, fe.srec_w)]
    plan = __import__("pynqs_amd").C_extension.plan_for(h1e, h2e, sorb, x.device).buf
    fe.run(x, plan, eps, 77, None)
    used = rec1[0] >= 0
    assert torch.equal(rec1[0], fe.srec_col) and torch.equal(rec1[1][used], fe.srec_w[used])
    fe.run(x, plan, eps, 78, None)
    assert not torch.equal(rec1[0], fe.srec_col)


def test_table_hits_and_capacity_overflow(fe2s2):
    """Determinants of the wave-function table take their amplitude from it inside the kernel (link <= -2) and stay out of the
    distinct list; a front end that is too small says so (and poisons the affected walkers) instead of dropping records silently."""
    from pynqs_amd import C_extension as cx, energy, public_function as pf, reduce_front as RF

    sorb, nele, noA, noB, n, eps = 40, 30, 15, 15, 96, 1e-2
    x = _dev(fe2s2["ci_space"][:n])
    h1e, h2e = _dev(fe2s2["h1e"]), _dev(fe2s2["h2e"])
    keys = _dev(fe2s2["ci_space"][:3000])
    g = torch.Generator().manual_seed(3)
    wf = (torch.rand(3000, generator=g, dtype=torch.float64) + 0.2).cuda()
    lut = pf.WavefunctionLUT(keys, wf, sorb, device=torch.device("cuda"))
    fe, nu = _front(x, h1e, h2e, sorb, noA, noB, eps, 500, lut.hashtable, seed=9)
    walker, col, w, link, onv, drawn = fe.records()
    pos, found = lut.find(onv)
    assert torch.equal(found, link <= -2) and torch.equal(pos[found], (-2 - link[found]).long())
    rows, own = _check_distinct(fe, nu, walker, onv, link, sorb)
    amp_u = (torch.rand(nu, generator=g, dtype=torch.float64) + 0.5).cuda()
    psi = torch.empty(onv.size(0), dtype=torch.float64, device="cuda")
    psi[own] = amp_u[rows]
    psi[found] = lut.wf_value[pos[found]]
    num = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, walker, w * psi)
    p0 = torch.zeros(n, dtype=torch.float64, device="cuda")
    p0[walker[col == 0]] = psi[col == 0]
    e, px = fe.contract(amp_u, lut.wf_value)
    assert torch.equal(px, p0)
    torch.testing.assert_close(e, num / p0, rtol=1e-10, atol=1e-12)
    # too small on purpose
    plan = cx.plan_for(h1e, h2e, sorb, x.device).buf
    small = RF.ReduceFrontEnd(n, sorb, nele, noA, noB, 0, torch.float64, x.device, cap_doubles=8, cap_unique=64)
    small.run(x, plan, 1e-4, 0, None)  # (hundreds of kept columns per segment)
    cnt = small.counters_host()
    assert small.overflowed(cnt) and cnt[1] & RF.OVERFLOW_DOUBLES and cnt[1] & RF.OVERFLOW_UNIQUE
    assert cnt[2] == int(small.seg_count[: small.nseg].max()) and cnt[2] > 8  # (what the segments NEEDED, not what fitted)
    e_bad, _ = small.contract(torch.ones(64, dtype=torch.float64, device="cuda"))
    assert bool(torch.isnan(e_bad).any())
    # reduce_front() grows the buffers by itself
    energy._FRONTS.clear()
    key_n = 7
    fe2, nu2 = energy.reduce_front(x[:key_n].contiguous(), h1e, h2e, sorb, nele, noA, noB, 1e-5, 0)
    assert not fe2.overflowed() and nu2 == int(torch.unique(fe2.records()[4], dim=0).size(0))


def test_float32_integrals():
    x, h1e, h2e, comb, hm = _setup(16, 4, 4, 20, seed=4, dtype=torch.float32)
    fe, nu = _front(x, h1e, h2e, 16, 4, 4, 0.3, 300, seed=5)
    walker, col, w, link, onv, drawn = fe.records()
    k = ~drawn
    keep = hm.abs() >= 0.3
    got = torch.zeros_like(keep)
    got[walker[k], col[k].long()] = True
    assert w.dtype == torch.float32 and torch.equal(got, keep) and torch.equal(w[k], hm[walker[k], col[k].long()])
    assert not keep[walker[drawn], col[drawn].long()].any()


@pytest.mark.parametrize("noA,noB,N", [(5, 7, 1025), (4, 4, 1000), (6, 2, 2500)])
def test_float32_integrals_with_draws(noA, noB, N):
    """float32 integrals through the round-4 kernel: the staging scratch of the enumeration is 4 KB then, less than the draws' arrays that
    lie over it afterwards (a layout that only counted the scratch let them run into the tile sums and the kept list: tools/fuzz_draws.py
    found drawn 'columns' that were the OR of two); also more than 1024 draws (the located columns wait in the draw slots' link words)."""
    x, h1e, h2e, comb, hm = _setup(16, noA, noB, 18, seed=9, dtype=torch.float32)
    eps = 0.45
    fe, nu = _front(x, h1e, h2e, 16, noA, noB, eps, N, seed=3)
    assert fe.row_f32 is not None
    walker, col, w, link, onv, drawn = fe.records()
    keep = hm.abs() >= eps
    k = ~drawn
    got = torch.zeros_like(keep)
    got[walker[k], col[k].long()] = True
    assert torch.equal(got, keep) and torch.equal(w[k], hm[walker[k], col[k].long()])
    dc, dw = col[drawn].long(), walker[drawn]
    assert int(dc.max()) < hm.size(1) and not keep[dw, dc].any() and torch.equal(onv[drawn], comb[dw, dc])
    S = torch.where(keep, torch.zeros_like(hm), hm.abs()).double().sum(1)
    hits = w[drawn].double().abs() * N / S[dw]
    assert float((hits - hits.round()).abs().max()) < 2e-2
    tot = torch.zeros(18, dtype=torch.float64, device=x.device).index_add_(0, dw, hits.round())
    assert torch.equal(tot, torch.full_like(tot, float(N)))
    flat = dw * hm.size(1) + dc
    assert bool((flat[1:] > flat[:-1]).all())


@pytest.mark.parametrize("N", [0, 300])
def test_step_is_graph_capturable(N, fe2s2):
    """reduce_front.ReduceStep: front end -> module on ALL rows of the distinct list -> contraction with static shapes, nothing read back;
    replayed from a HIP graph it gives what the eager step gives (same seed), and every replay draws afresh (the seed word lives in
    device memory and is bumped inside the graph)."""
    from pynqs_amd import C_extension as cx, reduce_front as RF
    from pynqs_amd.rbm import RealRBM

    d = golden("eloc_e2e_fe2s2.npz")
    sorb, nele, noA, noB, n, eps = 40, 30, 15, 15, 256, 1e-2
    x = _dev(fe2s2["ci_space"][:n])
    h1e, h2e = _dev(fe2s2["h1e"]), _dev(fe2s2["h2e"])
    plan = cx.plan_for(h1e, h2e, sorb, x.device).buf
    rbm = RealRBM(_dev(d["W"]), _dev(d["hb"]), _dev(d["vb"])).cuda().double()
    mk = lambda: RF.ReduceFrontEnd(n, sorb, nele, noA, noB, N, torch.float64, x.device, 256, 60000 + 300 * n * (N > 0), torch.float64)  # noqa: E731
    eager = RF.ReduceStep(mk(), plan, eps, rbm, seed=5, graph=False)
    graphed = RF.ReduceStep(mk(), plan, eps, rbm, seed=5, graph=True)
    e0, p0 = eager(x)
    g0, q0 = graphed(x)          # (two warm-up runs on a side stream bump the device seed before the capture: align the eager step)
    g0, q0 = g0.clone(), q0.clone()  # (a replay writes into the same output buffers)
    eager.check(); graphed.check()
    assert torch.equal(p0, q0) and bool(torch.isfinite(g0).all())
    if N == 0:
        torch.testing.assert_close(g0, e0, rtol=0, atol=1e-12)
        g1, _ = graphed(x)
        torch.testing.assert_close(g1, e0, rtol=0, atol=1e-12)
    else:
        eager.front.seed_dev.copy_(graphed.front.seed_dev - 1)   # the seed the last replay used
        e1, _ = eager(x)
        torch.testing.assert_close(g0, e1, rtol=0, atol=1e-10)   # same seed -> same draws -> same local energies
        first = graphed.front.srec_col.clone()
        g1, _ = graphed(x)
        assert not torch.equal(first, graphed.front.srec_col)    # the next replay draws other columns
        assert float((g1 - g0).abs().max()) > 0 and float((g1 - g0).abs().max()) < 5.0
        graphed.check()

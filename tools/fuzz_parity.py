"""Randomised differential test of the HIP kernels against the CPU oracle: random (sorb, noA, noB, batch, dtype) for the
drop-in kernel (bit-exact), the fused sample-space and RBM local energies (1e-8 Ha scaled) and the REDUCE compaction.
usage: python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from pynqs_amd import C_extension as cx, energy, public_function as pf, _native as N_

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda")
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def integrals(sorb):
    h1 = rng.random((sorb, sorb)) - 0.5
    h1 = (h1 + h1.T).reshape(-1)
    pair = sorb * (sorb - 1) // 2
    return h1, rng.random(pair * (pair + 1) // 2) - 0.5


def walkers(n, sorb, noA, noB):
    occ = np.zeros((n, sorb), dtype=np.uint8)
    for i in range(n):
        occ[i, 2 * rng.permutation(sorb // 2)[:noA]] = 1
        occ[i, 2 * rng.permutation(sorb // 2)[:noB] + 1] = 1
    return O.pm01_to_onv(occ, sorb)


t0, cases = time.time(), 0
last_report = t0
while time.time() - t0 < budget:
    sorb = int(rng.choice([4, 6, 8, 10, 12, 14, 16, 20, 24, 30, 36, 40, 66, 70, 130]))
    K = sorb // 2
    big = sorb > 40
    noA = int(rng.integers(0, min(K, 4 if big else K) + 1)); noB = int(rng.integers(0, min(K, 4 if big else K) + 1))
    if noA + noB == 0:
        continue
    n = int(rng.choice([1, 2, 3, 7, 33, 100])) if not big else int(rng.choice([1, 2, 5]))
    h1, h2 = integrals(sorb)
    x = walkers(n, sorb, noA, noB)
    nele = noA + noB
    for dt in (np.float32, np.float64):  # float64 last: ho / hm are reused below
        co, ho = O.comb_hij_fused(x, h1.astype(dt), h2.astype(dt), sorb, nele, noA, noB)
        comb, hm = cx.get_comb_hij_fused(G(x), G(h1.astype(dt)), G(h2.astype(dt)), sorb, nele, noA, noB)
        assert np.array_equal(comb.cpu().numpy(), co) and np.array_equal(hm.cpu().numpy(), ho), ("dropin", sorb, noA, noB, n, dt)
    h1e, h2e = G(h1), G(h2)
    scale = max(1.0, float(np.abs(ho).sum(1).max()))
    # REDUCE compaction == thresholded materialised row
    eps = float(rng.choice([0.0, 0.1, 0.3, 0.6]))
    row, col, onv, h, counts = energy.reduce_compact(G(x), h1e, h2e, sorb, nele, noA, noB, eps, sort=True)
    keep = torch.from_numpy(np.abs(ho) >= eps).to(dev)
    r2, c2 = torch.where(keep)
    assert torch.equal(row, r2) and torch.equal(col.long(), c2) and torch.equal(h, hm[keep]), ("reduce", sorb, noA, noB, n, eps)
    # fused RBM local energy
    H = int(rng.choice([1, 3, 8, 17, 40]))
    W = 0.2 * (rng.random((H, sorb)) - 0.5); hb = 2.0 * (rng.random(H) - 0.5); vb = 0.3 * (rng.random(sorb) - 0.5)
    e_ref, p_ref = O.eloc_simple_rbm(x, h1, h2, sorb, nele, noA, noB, W, hb, vb)
    e, p = cx.eloc_rbm(G(x), h1e, h2e, cx.RBMTable(G(W), G(hb), G(vb)), sorb, nele, noA, noB)
    assert np.allclose(p.cpu().numpy(), p_ref, rtol=1e-10, atol=0), ("rbm psi", sorb, noA, noB, n, H)
    assert np.abs(e.cpu().numpy() - e_ref).max() <= 1e-8 * max(1.0, np.abs(e_ref).max(), scale), ("rbm eloc", sorb, noA, noB, n, H)
    # the other real-parameter flavours (rbm.py:199-211) and the fixed-node Green's-function row, from the oracle's rows in numpy
    if co.shape[1] * n <= 400000:
        xs = O.onv_to_pm1(co.reshape(-1, co.shape[-1]), sorb)
        th = xs @ W.T + hb
        lncosh = (np.abs(th) + np.log1p(np.exp(-2.0 * np.abs(th)))).sum(1).reshape(n, -1)
        ax = (xs @ vb).reshape(n, -1)
        tab = cx.RBMTable(G(W), G(hb), G(vb))
        r_tanh = np.tanh(ax) / np.tanh(ax[:, :1]) * np.exp(lncosh - lncosh[:, :1])
        e_t, _ = cx.eloc_rbm(G(x), h1e, h2e, tab, sorb, nele, noA, noB, rbm_type="tanh")
        ok = np.abs(np.tanh(ax[:, 0])) > 1e-3  # (psi(x) ~ 0: the ratio is ill-conditioned in the reference too)
        assert np.abs(e_t.cpu().numpy() - (ho * r_tanh).sum(1))[ok].max(initial=0.0) <= 1e-8 * max(1.0, scale * np.abs(r_tanh[ok]).max(initial=0.0)), ("tanh", sorb, noA, noB, n, H)
        e_p, _ = cx.eloc_rbm(G(x), h1e, h2e, tab, sorb, nele, noA, noB, rbm_type="pRBM")
        r_ph = np.exp(1j * ((ax + lncosh) - (ax + lncosh)[:, :1]))
        assert np.abs(e_p.cpu().numpy() - (ho * r_ph).sum(1)).max() <= 1e-8 * scale, ("pRBM", sorb, noA, noB, n, H)
        # complex parameters (complex running products), and "cos" = the same kernel on i W, i b
        if N_.lib().pynqs_eloc_crbm_supported(sorb, nele, noA, noB, H):
            Wc = W + 1j * 0.2 * (rng.random((H, sorb)) - 0.5); hc = hb + 1j * (rng.random(H) - 0.5); vc = vb + 1j * 0.3 * (rng.random(sorb) - 0.5)
            thc = (xs @ Wc.T + hc).reshape(n, -1, H)
            axc = (xs @ vc).reshape(n, -1)
            r_c = np.exp(axc - axc[:, :1]) * np.prod(np.cosh(thc) / np.cosh(thc[:, :1]), axis=-1)
            e_c, _ = cx.eloc_crbm(G(x), h1e, h2e, cx.CRBMTable(G(Wc), G(hc), G(vc)), sorb, nele, noA, noB)
            assert np.abs(e_c.cpu().numpy() - (ho * r_c).sum(1)).max() <= 1e-8 * max(1.0, (np.abs(ho) * np.abs(r_c)).sum(1).max()), ("crbm", sorb, noA, noB, n, H)
        lam = float(np.median(ho[:, 0]))
        r = np.exp((ax + lncosh) - (ax + lncosh)[:, :1])
        keep = (ho < 0); keep[:, 0] = False
        v_sf = np.where(~keep, ho * r, 0.0)[:, 1:].sum(1)
        g_ref = np.where(keep, -ho * r, 0.0)
        g_ref[:, 0] = np.maximum(lam - ho[:, 0] - v_sf, 0.0)
        plan = cx.plan_for(h1e, h2e, sorb, dev)
        eg = torch.empty(n, dtype=torch.float64, device=dev); gk = torch.empty((n, co.shape[1]), dtype=torch.float64, device=dev)
        ng = torch.empty(n, dtype=torch.uint8, device=dev)
        N_.check(N_.lib().pynqs_green_rbm(G(x).data_ptr(), n, sorb, nele, noA, noB, plan.data_ptr(), tab.data_ptr(), tab.nhidden, 0, lam, eg.data_ptr(), None,
                                        gk.data_ptr(), ng.data_ptr(), torch.cuda.current_stream().cuda_stream), "green")
        assert np.abs(gk.cpu().numpy() - g_ref).max() <= 1e-8 * max(1.0, scale * np.abs(r).max()), ("green row", sorb, noA, noB, n, H)
        assert np.abs(eg.cpu().numpy() - e_ref).max() <= 1e-8 * max(1.0, np.abs(e_ref).max(), scale), ("green eloc", sorb, noA, noB, n, H)
        # the move from the column's rank == the reference's comb row
        u = G(rng.random((n, 1)))
        xi = torch.empty(n, dtype=torch.int64, device=dev); be = torch.empty((n, 1), dtype=torch.float64, device=dev)
        xn = torch.empty((n, (sorb - 1) // 64 + 1), dtype=torch.int64, device=dev)
        N_.check(N_.lib().pynqs_gfmc_sample_rank(gk.data_ptr(), n, u.data_ptr(), G(x).data_ptr(), sorb, nele, noA, noB, xi.data_ptr(), be.data_ptr(),
                                               xn.data_ptr(), torch.cuda.current_stream().cuda_stream), "sample_rank")
        if bool((be > 0).all()):
            picked = co[np.arange(n), xi.cpu().numpy()]
            assert np.array_equal(xn.cpu().numpy().view(np.uint64), np.ascontiguousarray(picked).view(np.uint64).reshape(n, -1)), ("rank move", sorb, noA, noB, n, xi.cpu().numpy()[:4])
    # get_hij_torch, 2-D mode with diagonal pairs (evaluated by whole waves)
    if n <= 33:
        kets = np.concatenate([x, co[0][: min(40, co.shape[1])]])
        hij = cx.get_hij_torch(G(x), G(kets), h1e, h2e, sorb, nele)
        assert np.array_equal(hij.cpu().numpy(), O.hij(x, kets, h1, h2, sorb, nele)), ("hij 2-D", sorb, noA, noB, n)
    # fused sample-space local energy: table = a random half of the connected determinants of walker 0 plus all walkers
    flat = np.unique(np.concatenate([co[0][rng.random(co.shape[1]) < 0.5], x]), axis=0)
    order = O.sort_keys(flat, sorb) if hasattr(O, "sort_keys") else None
    keys = flat[order] if order is not None else flat
    wf = rng.random(keys.shape[0]) + 0.1
    lut = pf.WavefunctionLUT(G(keys), G(wf), sorb, device=dev)
    el, _, psi0, _ = energy.local_energy(G(x), h1e, h2e, None, None, sorb, nele, noA, noB, WF_LUT=lut, use_sample_space=True)
    ks = lut.bra_key.cpu().numpy(); ws = lut.wf_value.cpu().numpy()
    e2, p2 = O.eloc_sample_space(x, h1, h2, sorb, nele, noA, noB, ks, ws)
    assert np.array_equal(psi0.cpu().numpy(), p2), ("ss psi0", sorb, noA, noB, n)
    assert np.abs(el.cpu().numpy() - e2).max() <= 1e-8 * max(1.0, np.abs(e2).max(), scale), ("ss eloc", sorb, noA, noB, n)
    # spin-flip-projected SAMPLE_SPACE: the fused partner-sum kernel against the materialising tensor path
    if co.shape[1] * n <= 400000:
        kf = torch.unique(torch.cat([lut.bra_key, pf.spin_flip_onv(lut.bra_key, sorb)]), dim=0)
        wfc = torch.from_numpy(rng.standard_normal(kf.size(0)) + 1j * rng.standard_normal(kf.size(0))).to(dev)
        lf = pf.WavefunctionLUT(kf, wfc, sorb, device=dev)
        pf.SpinProjection.init(nele, 0)
        res = []
        for fused in (True, False):
            energy.FUSED = fused
            try:
                ef, _, pz, _ = energy.local_energy(G(x), h1e, h2e, None, lambda x_, func: None, sorb, nele, noA, noB, WF_LUT=lf, use_sample_space=True,
                                                   dtype=torch.complex128, use_spin_flip=True, extra_norm=torch.tensor(1.1, dtype=torch.float64, device=dev))
            finally:
                energy.FUSED = True
            res.append((ef.cpu().numpy(), pz.cpu().numpy()))
        ok = np.isfinite(res[1][0])
        assert np.array_equal(res[0][1], res[1][1]), ("flip psi0", sorb, noA, noB, n)
        assert np.abs(res[0][0][ok] - res[1][0][ok]).max(initial=0.0) <= 1e-8 * max(1.0, np.abs(res[1][0][ok]).max(initial=0.0), scale), ("flip eloc", sorb, noA, noB, n)
    cases += 1
    if time.time() - last_report > 60:  # (a GPU run that stays silent for minutes is taken to be hung)
        last_report = time.time()
        print(f"... {cases} systems after {last_report - t0:.0f} s", flush=True)
print(f"fuzz ok: {cases} random systems in {time.time() - t0:.0f} s")

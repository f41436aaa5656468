"""pynqs_amd.energy (local_energy / total_energy / Func) on the GPU against outputs captured from the
reference's own Python (vmc/energy/eloc.py run on its CPU extension; tests/golden/eloc_e2e_fe2s2.npz).
Tolerance: 1e-8 Ha per determinant (the north-star bound); psi(x) to 1e-12 relative."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope="module")
def env(fe2s2):
    from pynqs_amd import energy, public_function as pf
    from pynqs_amd.rbm import RealRBM

    assert torch.cuda.is_available()
    d = golden("eloc_e2e_fe2s2.npz")
    dev = torch.device("cuda")
    f = fe2s2
    e = {
        "energy": energy, "pf": pf, "d": d, "dev": dev,
        "h1e": torch.from_numpy(f["h1e"]).to(dev), "h2e": torch.from_numpy(f["h2e"]).to(dev),
        "x": torch.from_numpy(d["x"]).to(dev),
        "rbm": RealRBM(torch.from_numpy(d["W"]), torch.from_numpy(d["hb"]), torch.from_numpy(d["vb"])).to(dev).double(),
    }
    torch.set_default_dtype(torch.float64)
    yield e
    torch.set_default_dtype(torch.float32)


def _ab(env, fp_batch=100000):
    pf = env["pf"]
    return lambda x, func: pf.ansatz_batch(func, x, fp_batch, 40, env["dev"], torch.double)


def _le(env, **kw):
    return env["energy"].local_energy(env["x"], env["h1e"], env["h2e"], env["rbm"], _ab(env), 40, 30, 15, 15, dtype=kw.pop("dtype", torch.double), **kw)


@pytest.mark.parametrize("fused", [True, False])
def test_simple_matches_reference_python(env, fused):
    """fused: the RealRBM ansatz takes the on-chip amplitude-ratio kernel (pynqs_eloc_rbm); else the generic
    materialise-and-forward path.  Both against the reference's Python."""
    d, energy = env["d"], env["energy"]
    old = energy.FUSED
    energy.FUSED = fused
    try:
        eloc, sloc, psi, times = _le(env, use_unique=True)
        np.testing.assert_allclose(psi.cpu().numpy(), d["psi_simple"], rtol=1e-12)
        np.testing.assert_allclose(eloc.cpu().numpy(), d["eloc_simple"], rtol=0, atol=TOL)
        assert float(sloc.abs().max()) == 0.0 and len(times) == 3
        e2, _, _, _ = _le(env, use_unique=False)
        np.testing.assert_allclose(e2.cpu().numpy(), d["eloc_simple"], rtol=0, atol=TOL)
    finally:
        energy.FUSED = old


@pytest.mark.parametrize("fused", [True, False])
def test_reduce_matches_reference_python(env, fused):
    d, energy = env["d"], env["energy"]
    old = energy.FUSED
    energy.FUSED = fused
    try:
        eloc, _, psi, _ = _le(env, reduce_psi=True, eps=1e-2, eps_sample=0)
        np.testing.assert_allclose(eloc.cpu().numpy(), d["eloc_reduce"], rtol=0, atol=TOL)
        np.testing.assert_allclose(psi.cpu().numpy(), d["psi_reduce"], rtol=1e-12)
        # LUT-assisted (Func: hit/miss split + unique on the misses)
        keys = torch.from_numpy(d["psi_lut_keys"][:512]).to(env["dev"])
        lut = env["pf"].WavefunctionLUT(keys, torch.from_numpy(d["psi_lut"][:512]).to(env["dev"]), 40, device=env["dev"])
        eloc, _, _, _ = _le(env, reduce_psi=True, eps=1e-2, eps_sample=0, WF_LUT=lut)
        np.testing.assert_allclose(eloc.cpu().numpy(), d["eloc_reduce_lut"], rtol=0, atol=TOL)
    finally:
        energy.FUSED = old


def test_reduce_compaction_is_exact(env):
    """Kept columns from the on-chip compaction == |Hmat| >= eps of the materialised matrix, same order."""
    from pynqs_amd import C_extension as cx

    energy = env["energy"]
    for eps in (1e-2, 0.0, 1e3):
        row, col, onv, h, counts = energy.reduce_compact(env["x"], env["h1e"], env["h2e"], 40, 30, 15, 15, eps, sort=True)
        # unsorted: the kernels' tile order -- same records, identical from run to run (no atomics)
        u1 = energy.reduce_compact(env["x"], env["h1e"], env["h2e"], 40, 30, 15, 15, eps)
        u2 = energy.reduce_compact(env["x"], env["h1e"], env["h2e"], 40, 30, 15, 15, eps)
        assert all(torch.equal(a, b) for a, b in zip(u1, u2))
        if col.numel():
            o = torch.argsort((u1[0] << 32) | u1[1].long())
            assert torch.equal(u1[1][o], col) and torch.equal(u1[3][o], h) and torch.equal(u1[2][o], onv)
        comb, hm = cx.get_comb_hij_fused(env["x"], env["h1e"], env["h2e"], 40, 30, 15, 15)
        keep = hm.abs() >= eps
        r2, c2 = torch.where(keep)
        assert torch.equal(row, r2) and torch.equal(col.long(), c2)
        assert torch.equal(h, hm[keep]) and torch.equal(onv, comb[keep])
        assert torch.equal(counts, keep.sum(1))


@pytest.mark.parametrize("fused", [True, "keys", False])
@pytest.mark.parametrize("cplx", [False, True])
def test_sample_space_matches_reference_python(env, fused, cplx):
    """fused True: the column-major kernel (excitation lists against the table), "keys": the key-major kernel (the table against the
    walkers), False: the reference's tensor algebra."""
    d, energy = env["d"], env["energy"]
    old_keys = energy.SS_KEYS
    energy.SS_KEYS = fused == "keys"
    fused = bool(fused)
    keys = torch.from_numpy(d["psi_lut_keys"]).to(env["dev"])
    wf = torch.from_numpy(d["psi_lut_c" if cplx else "psi_lut"]).to(env["dev"])
    lut = env["pf"].WavefunctionLUT(keys, wf, 40, device=env["dev"])
    old = energy.FUSED
    energy.FUSED = fused
    try:
        eloc, _, psi, _ = _le(env, WF_LUT=lut, use_sample_space=True, index=(0, 32), dtype=torch.complex128 if cplx else torch.double)
    finally:
        energy.FUSED = old
        energy.SS_KEYS = old_keys
    sfx = "_c" if cplx else ""
    np.testing.assert_allclose(eloc.cpu().numpy(), d["eloc_sample_space" + sfx], rtol=0, atol=TOL)
    np.testing.assert_allclose(psi.cpu().numpy(), d["psi_sample_space" + sfx], rtol=1e-14)


def test_total_energy_chunks_and_statistics(env):
    from pynqs_amd.stats import operator_statistics, dist_stats_onepass

    d, energy = env["d"], env["energy"]
    eloc, sloc, _ = energy.total_energy(env["x"], 5, 30000, env["h1e"], env["h2e"], env["rbm"], 40, 30, 15, 15, use_unique=True)
    np.testing.assert_allclose(eloc.cpu().numpy(), d["eloc_simple"], rtol=0, atol=TOL)
    prob = torch.from_numpy(d["prob"]).to(env["dev"])
    ref = torch.from_numpy(d["eloc_simple"]).to(env["dev"])
    st = operator_statistics(ref, prob, int(d["stat_counts"]), "E")
    for k in ("mean", "var", "sd", "se"):
        np.testing.assert_allclose(st[k].cpu().numpy(), d["stat_" + k], rtol=1e-13)
    m, v, sd, se = dist_stats_onepass(ref, prob, int(d["stat_counts"]), 1)
    np.testing.assert_allclose(m.cpu().numpy(), d["stat_mean"], rtol=1e-13)
    np.testing.assert_allclose(v.cpu().numpy(), d["stat_var"], rtol=1e-8)
    assert "<E> = " in repr(st)
    # the fused moments kernel (one launch instead of ~10): same numbers; complex and large inputs against torch
    from pynqs_amd.stats import dist_stats_moments

    m2, v2, sd2, se2 = dist_stats_moments(ref, prob, int(d["stat_counts"]), 1)
    np.testing.assert_allclose(m2.cpu().numpy(), d["stat_mean"], rtol=1e-13)
    np.testing.assert_allclose(v2.cpu().numpy(), d["stat_var"], rtol=1e-8)
    np.testing.assert_allclose(se2.cpu().numpy(), d["stat_se"], rtol=1e-8)
    g = torch.Generator(device=env["dev"]).manual_seed(5)
    for n in (1, 255, 70_001):
        z = torch.complex(torch.randn(n, generator=g, dtype=torch.float64, device=env["dev"]), torch.randn(n, generator=g, dtype=torch.float64, device=env["dev"]))
        p = torch.rand(n, generator=g, dtype=torch.float64, device=env["dev"]); p /= p.sum()
        a = dist_stats_moments(z, p, n, 1)
        b = dist_stats_onepass(z, p, n, 1)
        a2 = dist_stats_moments(z, p, n, 1)  # workspace reuse, bit-reproducible
        for u, w, u2 in zip(a, b, a2):
            np.testing.assert_allclose(u.cpu().numpy(), w.cpu().numpy(), rtol=1e-10, atol=1e-13)
            assert torch.equal(u, u2)


def test_total_energy_sizes_its_own_chunks(env):
    """nbatch = 0: chunks sized for the fused path that runs (public_function.get_nbatch(fused=...)); same numbers as one call."""
    energy, pf, d = env["energy"], env["pf"], env["d"]
    keys = torch.from_numpy(d["psi_lut_keys"]).to(env["dev"])
    lut = pf.WavefunctionLUT(keys, torch.from_numpy(d["psi_lut"]).to(env["dev"]), 40, device=env["dev"])
    args = (env["h1e"], env["h2e"], env["rbm"], 40, 30, 15, 15)
    for kw in (dict(WF_LUT=lut, use_sample_space=True), dict(reduce_psi=True, eps=1e-2), dict()):
        a, _, _ = energy.total_energy(env["x"], 0, 100000, *args, **kw)
        b, _, _ = energy.total_energy(env["x"], -1, 100000, *args, **kw)
        torch.testing.assert_close(a, b, rtol=0, atol=1e-10)  # (the key-major sample-space kernel adds with atomics: last-bit differences)
    assert energy.auto_nbatch(env["x"], env["h1e"], 40, 30, 15, 15, env["rbm"], lut, torch.double, False, 0, True, False, False, False) == env["x"].size(0)


@pytest.mark.parametrize("eps_sample", [0, 150])
def test_total_energy_look_ahead_on_a_second_stream(env, eps_sample):
    """REDUCE over several chunks of walkers: the front end of chunk k + 1 is enqueued on a second stream while the ansatz works on chunk k
    (SURVEY 7.6); same numbers as everything on one stream, also with draws (the seeds are drawn in chunk order either way)."""
    energy = env["energy"]
    args = (env["h1e"], env["h2e"], env["rbm"], 40, 30, 15, 15)
    out = {}
    old_rbm, energy.FUSED_RBM = energy.FUSED_RBM, False  # (the module path: the ansatz runs on the main stream)
    try:
        for ov in (True, False):
            old, energy.OVERLAP = energy.OVERLAP, ov
            try:
                torch.manual_seed(99)
                out[ov] = energy.total_energy(env["x"], 5, 100000, *args, reduce_psi=True, eps=1e-2, eps_sample=eps_sample)[0]
            finally:
                energy.OVERLAP = old
    finally:
        energy.FUSED_RBM = old_rbm
    assert torch.equal(out[True], out[False])
    if eps_sample == 0:
        np.testing.assert_allclose(out[True].cpu().numpy(), env["d"]["eloc_reduce"], rtol=0, atol=TOL)


def test_total_energy_look_ahead_with_a_non_contiguous_batch(env):
    """The walkers as a strided view (every other row of a larger tensor) under the look-ahead: the chunks are made contiguous once, on the
    main stream, before anything is enqueued on the second one (a per-chunk copy on the main stream raced with the side stream's kernel);
    the small buffers force the overflow / regrow path of a look-ahead workspace as well."""
    energy = env["energy"]
    args = (env["h1e"], env["h2e"], env["rbm"], 40, 30, 15, 15)
    x = env["x"]
    wide = torch.zeros((2 * x.size(0), x.size(1)), dtype=torch.uint8, device=x.device)
    wide[0::2] = x
    wide[1::2] = x.flip(0)
    xv = wide[0::2]
    assert not xv.is_contiguous() and torch.equal(xv, x)
    old_rbm, energy.FUSED_RBM = energy.FUSED_RBM, False
    try:
        energy._FRONTS.clear()
        want = energy.total_energy(x, 5, 100000, *args, reduce_psi=True, eps=1e-2)[0]
        for _ in range(3):  # (first pass: fresh workspaces that overflow and are regrown; then the cached ones)
            got = energy.total_energy(xv, 5, 100000, *args, reduce_psi=True, eps=1e-2)[0]
            assert torch.equal(got, want)
        energy._FRONTS.clear()
        got = energy.total_energy(xv, 5, 100000, *args, reduce_psi=True, eps=1e-4)[0]   # (hundreds of kept columns: every first launch overflows)
        want4 = energy.total_energy(x, -1, 100000, *args, reduce_psi=True, eps=1e-4)[0]
        assert float((got - want4).abs().max()) < 1e-10
    finally:
        energy.FUSED_RBM = old_rbm
    np.testing.assert_allclose(want.cpu().numpy(), env["d"]["eloc_reduce"], rtol=0, atol=TOL)


def test_spin_flip_helpers_and_eps0_consistency(env):
    """Helper forms (packed / occupation rows) agree, and REDUCE with eps = 0 equals SIMPLE for the projected form.  (Parity of the
    projected and multi-psi local energies themselves against the reference's Python: test_gpu_energy_flip.py.)"""
    from pynqs_amd import C_extension as cx

    pf, energy = env["pf"], env["energy"]
    pf.SpinProjection.init(30, 0)
    x = env["x"][:6].contiguous()
    ab = _ab(env)
    en = torch.tensor(1.3, dtype=torch.float64, device=env["dev"])
    eloc, _, psi, _ = energy.local_energy(x, env["h1e"], env["h2e"], env["rbm"], ab, 40, 30, 15, 15, use_spin_flip=True, extra_norm=en)
    comb, hm = cx.get_comb_hij_fused(x, env["h1e"], env["h2e"], 40, 30, 15, 15)
    flat = comb.reshape(-1, 8)
    with torch.no_grad():
        p = ab(flat, env["rbm"]).reshape(6, -1)
        pfl = ab(pf.spin_flip_onv(flat, 40), env["rbm"]).reshape(6, -1)
    eta_m = pf.spin_flip_sign(flat, 40).reshape(6, -1)
    f_psi = (p + pf.SpinProjection.eta * eta_m * pfl) / en**2
    want = ((f_psi.T / p[:, 0]).T * hm).sum(-1)
    np.testing.assert_allclose(eloc.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-10)
    # uint8 and +-1 forms of the spin-flip helpers agree
    pm = cx.onv_to_tensor(flat[:500].contiguous(), 40)
    assert torch.equal(pf.spin_flip_sign(flat[:500], 40), pf.spin_flip_sign(((pm + 1) / 2).to(torch.int64), 40))
    assert torch.equal(cx.onv_to_tensor(pf.spin_flip_onv(flat[:500], 40).contiguous(), 40), pf.spin_flip_onv(pm, 40))
    # REDUCE + spin flip == SIMPLE + spin flip when nothing is filtered (eps = 0)
    e2, _, _, _ = energy.local_energy(x, env["h1e"], env["h2e"], env["rbm"], ab, 40, 30, 15, 15, use_spin_flip=True, extra_norm=en,
                                      reduce_psi=True, eps=0.0)
    np.testing.assert_allclose(e2.cpu().numpy(), eloc.cpu().numpy(), rtol=0, atol=1e-10)


def test_spin_flip_rand_reaches_exactly_the_sd_space(env):
    """Like cpp_src/test/test-spin-flip.py: the set of proposed states == the rows of get_comb_tensor, and the
    move index is uniform on [0, ncomb]."""
    from pynqs_amd import C_extension as cx

    for (sorb, noA, noB) in [(8, 2, 2), (12, 3, 2), (130, 2, 1)]:
        L = (sorb - 1) // 64 + 1
        occ = torch.zeros(sorb, dtype=torch.uint8)
        occ[[2 * i for i in range(noA)]] = 1; occ[[2 * i + 1 + 2 * (i % 2) for i in range(noB)]] = 1
        x0 = cx.tensor_to_onv(occ.cuda(), sorb)
        comb, _ = cx.get_comb_tensor(x0, sorb, noA + noB, noA, noB)
        ncomb = comb.size(1)
        n = 400 * ncomb
        pm, new = cx.spin_flip_rand(x0.repeat(n, 1).contiguous(), sorb, noA + noB, noA, noB, seed=1234)
        assert new.shape == (n, 8 * L) and torch.equal(pm, cx.onv_to_tensor(new, sorb))
        uniq, counts = torch.unique(new, dim=0, return_counts=True)
        assert torch.equal(uniq, torch.unique(comb[0], dim=0))
        # x itself is produced by r0 = 0 only: expected share 1/(nsd+1) each (chi-square style bound)
        exp = n / ncomb
        assert float(((counts - exp) ** 2 / exp).sum()) < 2.0 * ncomb
        pm2, new2 = cx.spin_flip_rand(x0.repeat(n, 1).contiguous(), sorb, noA + noB, noA, noB, seed=1234)
        assert not torch.equal(new, new2)  # a fresh stream on every call


def test_gfmc_step_matches_direct_tensor_algebra(env):
    from pynqs_amd import C_extension as cx, gfmc

    x = env["x"][:5].contiguous()
    ab = _ab(env)
    eloc, gk, comb, stop, neg = gfmc.green_kernel(x, -100.0, env["h1e"], env["h2e"], env["rbm"], ab, 40, 30, 15, 15)
    comb2, hm = cx.get_comb_hij_fused(x, env["h1e"], env["h2e"], 40, 30, 15, 15)
    with torch.no_grad():
        psi = ab(comb2.reshape(-1, 8), env["rbm"]).reshape(5, -1)
    ratio = psi / psi[:, :1]
    keep = (torch.sign(hm) * torch.sign(ratio)) < 0  # cos(alpha + gamma) < 0 for real psi
    keep[:, 0] = True
    heff = torch.where(keep, hm, 0.0)
    heff[:, 0] += (torch.where(~keep, hm, 0.0) * ratio).sum(-1)
    want_e = (ratio * heff).sum(-1)
    np.testing.assert_allclose(eloc.cpu().numpy(), want_e.cpu().numpy(), rtol=0, atol=1e-9)
    K = -heff; K[:, 0] += -100.0
    np.testing.assert_allclose(gk.cpu().numpy(), torch.clamp(ratio * K, min=0.0).cpu().numpy() if bool(neg.any()) else (ratio * K).cpu().numpy(),
                               rtol=1e-12, atol=1e-12)
    assert isinstance(comb, gfmc.CombRows)  # (real RBM trial function: the fused row, nothing materialised)
    assert torch.equal(comb.materialize(), comb2) and stop is False
    # the fixed-node E_loc equals the plain E_loc when psi has no sign structure issue: sum_k ratio*H
    r = torch.full((5, 1), 0.37, dtype=torch.float64, device=x.device)
    x_new, w_new, beta, acc = gfmc.sample_update(x, torch.ones(5, dtype=torch.float64, device=x.device), comb, gk, r)
    cum = gk.cumsum(-1) / gk.sum(-1, keepdim=True)
    idx = (cum < 0.37).sum(-1)
    assert torch.equal(x_new, comb2[torch.arange(5), idx]) and torch.allclose(w_new, gk.sum(-1))


@pytest.mark.parametrize("n,m,L", [(7, 1, 1), (5, 27, 1), (33, 255, 1), (9, 256, 2), (9, 257, 3), (64, 7876, 1), (3, 300_001, 2)])
def test_gfmc_sample_kernel_matches_sequential_oracle(n, m, L):
    """pynqs_gfmc_sample (sum + cumsum + searchsorted + gather in one kernel) against a sequential float64 cumsum:
    index = first k with cum[k] >= u * beta (gfmc/walker.py:268-271); a different k is accepted only when the
    running sum is within rounding of the target there.  Rows contain exact zeros, one row is a single spike and
    the uniforms include 0 and 1 - 2^-53."""
    from pynqs_amd import gfmc

    g = torch.Generator().manual_seed(100 * n + m)
    gk = torch.rand(n, m, generator=g, dtype=torch.float64)
    gk[torch.rand(n, m, generator=g) < 0.6] = 0.0
    gk[0] = 0.0
    gk[0, m // 2] = 3.0
    gk[:, 0] += 1e-3  # the diagonal (Lambda - H_00) is positive
    u = torch.rand(n, 1, generator=g, dtype=torch.float64)
    u[0, 0] = 0.0
    if n > 1:
        u[1, 0] = 1.0 - 2.0 ** -53
    comb = torch.randint(0, 256, (n, m, 8 * L), generator=g, dtype=torch.uint8)
    w = torch.rand(n, generator=g, dtype=torch.float64)
    d = torch.device("cuda")
    old = gfmc.FUSED_SAMPLE
    try:
        gfmc.FUSED_SAMPLE = True
        x_new, w_new, beta, acc = gfmc.sample_update(None, w.to(d), comb.to(d), gk.to(d), u.to(d))
    finally:
        gfmc.FUSED_SAMPLE = old
    cum = np.cumsum(gk.numpy(), axis=1)
    tot = cum[:, -1]
    np.testing.assert_allclose(beta.cpu().numpy().ravel(), tot, rtol=1e-13)
    np.testing.assert_allclose(w_new.cpu().numpy(), w.numpy() * tot, rtol=1e-13)
    xn = x_new.cpu()
    nz = 0
    for i in range(n):
        hits = np.nonzero((comb[i] == xn[i]).all(dim=1).numpy())[0]
        assert hits.size >= 1
        target = u[i, 0].item() * tot[i]
        want = min(int(np.searchsorted(cum[i], target, side="left")), m - 1)
        if want not in hits:
            k = int(hits[0])
            lo = cum[i, k - 1] if k else 0.0
            assert abs(cum[i, min(k, want)] - target) <= 1e-12 * tot[i] or (lo - 1e-12 * tot[i] <= target <= cum[i, k] + 1e-12 * tot[i]), (i, k, want)
        nz += int(want != 0)
    assert abs(acc - nz) <= 1


def test_reduce_compaction_large_system():
    """sorb 160 with 40 + 40 electrons, 3.8 M columns per walker, rows cut into chunks over many workgroups: the
    compaction (atomically reserved records + sort) must equal |Hmat| >= eps of the materialised row, in order."""
    from conftest import rand_occ, synth_integrals
    from pynqs_amd import C_extension as cx, energy

    sorb, no = 160, 40
    h1, h2 = synth_integrals(sorb)
    d = torch.device("cuda")
    h1e, h2e = torch.from_numpy(h1).to(d), torch.from_numpy(h2).to(d)
    x = cx.tensor_to_onv(torch.from_numpy(rand_occ(2, sorb, no, no, seed=9)).to(d), sorb)
    row, col, onv, h, counts = energy.reduce_compact(x, h1e, h2e, sorb, 2 * no, no, no, 0.49, sort=True)
    comb, hm = cx.get_comb_hij_fused(x, h1e, h2e, sorb, 2 * no, no, no)
    keep = hm.abs() >= 0.49
    r2, c2 = torch.where(keep)
    assert torch.equal(row, r2) and torch.equal(col.long(), c2)
    assert torch.equal(h, hm[keep]) and torch.equal(onv, comb[keep])
    assert torch.equal(counts, keep.sum(1))


def _random_walkers(rng, n, sorb, no):
    k = sorb // 2
    L = (sorb - 1) // 64 + 1
    words = np.zeros((n, L), dtype=np.uint64)
    for spin in (0, 1):
        orb = 2 * np.argsort(rng.random((n, k)), axis=1)[:, :no] + spin
        for c in range(no):
            o = orb[:, c]
            np.bitwise_or.at(words, (np.arange(n), o // 64), np.uint64(1) << (o % 64).astype(np.uint64))
    return words


def _excite(rng, words, sorb, count, singles):
    """`count` determinants, each a single (one spin) or an alpha-beta double excitation of one of `words`."""
    n = words.shape[0]
    src = words[np.arange(count) % n].copy()
    orb = np.arange(sorb)
    occ = ((src[:, orb // 64] >> (orb % 64).astype(np.uint64)) & np.uint64(1)).astype(bool)
    rows = np.arange(count)
    for spin in ((0,) if singles else (0, 1)):
        same = (orb % 2 == spin)[None, :]
        score = rng.random((count, sorb))
        for o in (np.argmax(np.where(occ & same, score, -1.0), axis=1), np.argmax(np.where(~occ & same, score, -1.0), axis=1)):
            src[rows, o // 64] ^= np.uint64(1) << (o % 64).astype(np.uint64)
    return src


@pytest.mark.parametrize("sorb,no,nkeys,use_hash", [(40, 5, 150, True), (40, 5, 200, True), (72, 6, 150, True), (72, 6, 200, True),
                                                    (136, 4, 3000, True), (136, 4, 5000, True), (40, 5, 300_000, True),
                                                    (40, 5, 200, False), (136, 4, 3000, False)])
@pytest.mark.parametrize("key_major", [False, True])
def test_sample_space_kernel_filter_levels(sorb, no, nkeys, use_hash, key_major):
    """The fused SAMPLE_SPACE kernel with its candidate filters (Zobrist hash in LDS; second level in global memory
    when the first has < 6 bits per key: 200 and 5000 keys here, 150 and 3000 keys take the one-level kernel), without
    them, and with the sorted-key search, against the oracle, on sample spaces that hold the walkers, singles and doubles of them, and unrelated determinants."""
    from oracle import oracle as O
    from pynqs_amd import energy, public_function as pf

    rng = np.random.default_rng(1000 * sorb + nkeys)
    dev = torch.device("cuda")
    n = 24
    x = _random_walkers(rng, n, sorb, no)
    pool = np.concatenate([x, _excite(rng, x, sorb, nkeys // 3, True), _excite(rng, x, sorb, nkeys // 3, False),
                           _random_walkers(rng, nkeys, sorb, no)])
    keys = np.unique(pool, axis=0)
    keys = keys[rng.permutation(keys.shape[0])[:nkeys]]
    keys = np.unique(np.concatenate([x, keys]), axis=0)
    wf = rng.standard_normal(keys.shape[0]) + 1j * rng.standard_normal(keys.shape[0])
    h1 = rng.standard_normal((sorb, sorb)); h1 = (h1 + h1.T).reshape(-1)
    pair = sorb * (sorb - 1) // 2
    h2 = rng.standard_normal(pair * (pair + 1) // 2)
    L = x.shape[1]
    tb = lambda w: torch.from_numpy(w.view(np.uint8).reshape(-1, 8 * L)).to(dev)
    # 300 000 keys: more than the LDS filter takes (< 1 bit per key) -> the unfiltered hash kernel; use_hash False: the
    # binary search over the sorted keys, as the reference does
    old_flag = pf.USE_HASH
    pf.USE_HASH = use_hash
    try:
        lut = pf.WavefunctionLUT(tb(keys), torch.from_numpy(wf).to(dev), sorb, device=dev)
    finally:
        pf.USE_HASH = old_flag
    assert (lut.hashtable is not None) == use_hash
    old_keys, energy.SS_KEYS = energy.SS_KEYS, key_major  # the column-major kernels with their filters, or the key-major kernel on the same tables
    try:
        e, _, p0, _ = energy.local_energy(tb(x), torch.from_numpy(h1).to(dev), torch.from_numpy(h2).to(dev), None, None, sorb, 2 * no, no, no,
                                          WF_LUT=lut, use_sample_space=True, dtype=torch.complex128)
    finally:
        energy.SS_KEYS = old_keys
    e_ref, p_ref = O.eloc_sample_space(x.view(np.uint8).reshape(n, 8 * L), h1, h2, sorb, 2 * no, no, no, lut.bra_key.cpu().numpy(),
                                       lut.wf_value.cpu().numpy())
    np.testing.assert_array_equal(p0.cpu().numpy(), p_ref)
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=0, atol=TOL, err_msg=f"|E_loc|max = {float(np.abs(e_ref).max()):.6g} Ha")


@pytest.mark.parametrize("L,n,distinct", [(1, 1, 1), (1, 1000, 7), (1, 700_000, 90_000), (2, 300_000, 300_000), (3, 250_000, 1000)])
def test_unique_onv_on_gpu(L, n, distinct):
    """public_function.unique_onv on the GPU (pynqs_unique_first: hash table of row indices, no sort) against
    torch.unique(dim=0): same set of rows, rows[inverse] reproduces the input, unique rows in order of first appearance,
    identical from run to run."""
    from pynqs_amd.public_function import unique_onv

    g = torch.Generator().manual_seed(100 * L + n % 97)
    pool = torch.randint(-2**62, 2**62, (distinct, L), generator=g, dtype=torch.int64)
    pool[:, -1] &= (1 << 40) - 1  # stay within 64 (L - 1) + 40 orbitals
    x = pool[torch.randint(0, distinct, (n,), generator=g)].contiguous().view(torch.uint8).view(n, 8 * L).cuda()
    u, inv = unique_onv(x)
    assert torch.equal(u[inv], x)
    ref = torch.unique(x, dim=0)
    assert u.size(0) == ref.size(0)
    assert torch.equal(torch.unique(u, dim=0), ref)
    # order of first appearance
    firsts = torch.full((u.size(0),), n, dtype=torch.int64, device=x.device).scatter_reduce(0, inv, torch.arange(n, device=x.device), "amin")
    assert bool((firsts[1:] > firsts[:-1]).all())
    u2, inv2 = unique_onv(x)
    assert torch.equal(u, u2) and torch.equal(inv, inv2)


def test_float32_inputs_take_the_fused_kernels(env, monkeypatch):
    """float32 integrals / amplitudes (the reference dispatches f32 and f64 everywhere, cpu_tensor.cpp:249,298): the fused
    SAMPLE_SPACE and SIMPLE-with-RBM kernels run on an exact float64 up-conversion -- results within float32 rounding of the
    float64 goldens, returned in float32, nothing materialised."""
    from pynqs_amd.rbm import RealRBM

    d, energy, pf, dev = env["d"], env["energy"], env["pf"], env["dev"]
    h1f, h2f = env["h1e"].float(), env["h2e"].float()
    monkeypatch.setattr(energy, "get_comb_hij_fused", lambda *a, **k: (_ for _ in ()).throw(AssertionError("materialising path taken")))
    # SAMPLE_SPACE, complex64 table
    keys = torch.from_numpy(d["psi_lut_keys"]).to(dev)
    wf = torch.from_numpy(d["psi_lut_c"]).to(dev).to(torch.complex64)
    lut = pf.WavefunctionLUT(keys, wf, 40, device=dev)
    e, _, p0, _ = energy.local_energy(env["x"], h1f, h2f, None, lambda x_, func: None, 40, 30, 15, 15, dtype=torch.complex64, WF_LUT=lut,
                                      use_sample_space=True, index=(0, 32))
    assert e.dtype == torch.complex64 and p0.dtype == torch.complex64
    np.testing.assert_allclose(e.cpu().numpy(), d["eloc_sample_space_c"], rtol=2e-5, atol=2e-4)
    # SIMPLE with a float32 RBM
    torch.set_default_dtype(torch.float32)
    try:
        rbm32 = RealRBM(torch.from_numpy(d["W"]).float(), torch.from_numpy(d["hb"]).float(), torch.from_numpy(d["vb"]).float()).to(dev)
        ab = lambda x, func: pf.ansatz_batch(func, x, 100000, 40, dev, torch.float32)  # noqa: E731
        e, _, p0, _ = energy.local_energy(env["x"], h1f, h2f, rbm32, ab, 40, 30, 15, 15, dtype=torch.float32)
    finally:
        torch.set_default_dtype(torch.float64)
    assert e.dtype == torch.float32
    np.testing.assert_allclose(e.cpu().numpy(), d["eloc_simple"], rtol=2e-5, atol=2e-4)
    np.testing.assert_allclose(p0.cpu().numpy(), d["psi_simple"], rtol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("eps_sample", [0, 150])
def test_rbm_amplitudes_enqueued_before_the_counters_are_read(env, eps_sample, monkeypatch):
    """local_energy (REDUCE) with an RBM of the reference's family: amplitude kernel and contraction enqueued behind the front end before
    its counters are waited for (energy._rbm_ahead) -- the same numbers as the wait-then-launch order (1e-10 Ha), also on the first call of
    a shape (whose buffers overflow and are regrown: the speculative result is dropped) and inside total_energy's look-ahead."""
    energy = env["energy"]
    kw = dict(reduce_psi=True, eps=1e-2, eps_sample=eps_sample)
    out = {}
    for spec in (False, True):
        monkeypatch.setattr(energy, "SPECULATE_RBM", spec)
        energy.reset_caches()
        calls = {"n": 0}
        ahead = energy._rbm_ahead
        monkeypatch.setattr(energy, "_rbm_ahead", lambda *a, **k: (calls.__setitem__("n", calls["n"] + 1), ahead(*a, **k))[1])
        res = []
        for _ in range(3):   # (first call: sizing / overflow; later calls: the cached workspace)
            torch.manual_seed(11)
            res.append(_le(env, **kw)[0])
        torch.manual_seed(11)
        res.append(energy.total_energy(env["x"], 17, -1, env["h1e"], env["h2e"], env["rbm"], 40, 30, 15, 15, reduce_psi=True, eps=1e-2, eps_sample=eps_sample)[0])
        monkeypatch.setattr(energy, "_rbm_ahead", ahead)
        assert (calls["n"] > 0) == spec
        out[spec] = res
    # (not bit for bit: which walker becomes the PARENT of a distinct x' -- the first to insert it -- depends on the workgroups' timing, and
    # psi(x') from another parent differs in the last bits; the draws' seeds come from torch's generator: the same sequence of calls draws
    # the same seeds)
    for a, b in zip(out[False], out[True]):
        assert torch.isfinite(a).all() and float((a - b).abs().max()) < 1e-10
    assert float((out[True][0] - out[True][2]).abs().max()) < 1e-10

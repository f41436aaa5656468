"""The kernels at BASELINE.json's shapes (half filling): C2 sorb 56 (7a7b) with 4096 walkers,
C3/C4 at the stated size sorb 120 (30a30b, two ONV words) and C5 sorb 184 (46a46b, three words) -- chunked rows, 1024-thread
workgroups, windowed RBM, non-temporal stores: the configurations the bench runs, against the CPU oracle (bit-exact for
comb / Hmat, 1e-8 Ha relative to the row's magnitude for local energies on the dense synthetic integrals) and through
size-independent properties on the full batch."""
import numpy as np
import pytest
import torch

from conftest import rand_occ, synth_integrals

pytestmark = pytest.mark.gpu
TOL = 1e-8


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _popcounts(comb_u8: torch.Tensor):
    """(alpha, beta) electron counts of every packed determinant: even / odd bits of each byte."""
    lut_a = torch.tensor([bin(b & 0x55).count("1") for b in range(256)], dtype=torch.int32, device=comb_u8.device)
    lut_b = torch.tensor([bin(b & 0xAA).count("1") for b in range(256)], dtype=torch.int32, device=comb_u8.device)
    idx = comb_u8.long()
    return lut_a[idx].sum(-1), lut_b[idx].sum(-1)


def test_c2_sorb56_4096_walkers():
    from oracle import oracle as O
    from pynqs_amd import C_extension as cx

    sorb, no, n = 56, 7, 4096
    h1, h2 = synth_integrals(sorb)
    onv_np = O.pm01_to_onv(rand_occ(n, sorb, no, no, seed=4321), sorb)
    x, h1e, h2e = G(onv_np), G(h1), G(h2)
    comb, hm = cx.get_comb_hij_fused(x, h1e, h2e, sorb, 2 * no, no, no)
    assert comb.shape == (n, 30724, 8) and hm.shape == (n, 30724)
    # oracle on a slice spread over the batch (first / last workgroups included)
    sl = np.r_[0:4096:137, 4095]
    co, ho = O.comb_hij_fused(onv_np[sl], h1, h2, sorb, 2 * no, no, no)
    idx = torch.from_numpy(sl).cuda()
    assert np.array_equal(comb[idx].cpu().numpy(), co) and np.array_equal(hm[idx].cpu().numpy(), ho)
    # properties on the full batch: column 0 is x, particle numbers conserved per spin, kets of a row distinct from x,
    # chunk invariance (the same rows whatever the batch they are launched in)
    assert torch.equal(comb[:, 0], x)
    for b in range(0, n, 512):
        na, nb = _popcounts(comb[b:b + 512])
        assert bool((na == no).all()) and bool((nb == no).all())
        assert bool((comb[b:b + 512, 1:] != x[b:b + 512, None]).any(-1).all())
    c2, h2_ = cx.get_comb_hij_fused(x[1000:1003].contiguous(), h1e, h2e, sorb, 2 * no, no, no)
    assert torch.equal(c2, comb[1000:1003]) and torch.equal(h2_, hm[1000:1003])
    # Hermiticity through the generic pair kernel: <x|H|x'> == <x'|H|x> for sampled columns
    cols = torch.randint(1, 30724, (64,), generator=torch.Generator().manual_seed(1)).cuda()
    kets = comb[7, cols].contiguous()
    back = cx.get_hij_torch(kets, x[7:8].contiguous(), h1e, h2e, sorb, 2 * no)  # [64, 1]
    assert torch.equal(back[:, 0], hm[7, cols])
    del comb, hm
    # fused SIMPLE local energy with the RBM in the kernel (alpha = 2), 64 walkers against the oracle
    g = np.random.default_rng(7)
    H = 2 * sorb
    W, hb, vb = 0.01 * (g.random((H, sorb)) - 0.5), 0.01 * (g.random(H) - 0.5), 0.1 * (g.random(sorb) - 0.5)
    e_ref, p_ref = O.eloc_simple_rbm(onv_np[:64], h1, h2, sorb, 2 * no, no, no, W, hb, vb)
    e, p = cx.eloc_rbm(x[:64].contiguous(), h1e, h2e, cx.RBMTable(G(W), G(hb), G(vb)), sorb, 2 * no, no, no)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref, rtol=1e-11)
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=0, atol=TOL, err_msg=f"|E_loc|max = {float(np.abs(e_ref).max()):.6g} Ha")


def _rbm_eloc_logdomain(O, comb_row, hm_row, sorb, W, hb, vb, chunk=1 << 19):
    """E_loc and psi(x) of ONE walker from the ORACLE's materialised row (kets and matrix elements), with the amplitudes
    from a plain float64 PyTorch evaluation of ln psi = a.x + sum_h ln 2cosh(theta_h) in chunks of columns (the oracle's
    scalar forward needs minutes per walker at sorb 184 x 368 hidden units: 2.4e9 transcendentals)."""
    Wt, hbt, vbt = G(W), G(hb), G(vb)
    parts = []
    for b in range(0, comb_row.shape[0], chunk):
        xs = G(O.onv_to_pm1(comb_row[b:b + chunk], sorb))
        th = (xs @ Wt.T + hbt).abs()
        parts.append(xs @ vbt + (th + torch.log1p(torch.exp(-2.0 * th))).sum(1))
    lnpsi = torch.cat(parts)
    e = (G(hm_row) * torch.exp(lnpsi - lnpsi[0])).sum()
    return float(e), float(torch.exp(lnpsi[0]))


@pytest.mark.parametrize("sorb,no,n,H", [(120, 30, 3, 240), (184, 46, 2, 368)])
def test_half_filled_multiword(sorb, no, n, H):
    from oracle import oracle as O
    from pynqs_amd import C_extension as cx, energy, public_function as pf

    L = (sorb - 1) // 64 + 1
    h1, h2 = synth_integrals(sorb)
    onv_np = O.pm01_to_onv(rand_occ(n, sorb, no, no, seed=4321), sorb)
    x, h1e, h2e = G(onv_np), G(h1), G(h2)
    # ---- drop-in rows: bit-exact, f64 (all walkers) and f32 (one walker) ----------------------------------------
    co, ho = O.comb_hij_fused(onv_np, h1, h2, sorb, 2 * no, no, no)
    comb, hm = cx.get_comb_hij_fused(x, h1e, h2e, sorb, 2 * no, no, no)
    assert comb.shape == co.shape and np.array_equal(comb.cpu().numpy(), co)
    assert np.array_equal(hm.cpu().numpy(), ho)
    _, ho32 = O.comb_hij_fused(onv_np[:1], h1.astype(np.float32), h2.astype(np.float32), sorb, 2 * no, no, no)
    _, hm32 = cx.get_comb_hij_fused(x[:1].contiguous(), h1e.float(), h2e.float(), sorb, 2 * no, no, no)
    assert np.array_equal(hm32.cpu().numpy(), ho32)
    del hm32
    # ---- REDUCE compaction: the kept columns are |Hmat| >= eps of the oracle's row, same order after the sort ----
    eps = 0.49
    row, col, onv, h, counts = energy.reduce_compact(x, h1e, h2e, sorb, 2 * no, no, no, eps, sort=True)
    keep = np.abs(ho) >= eps
    r2, c2 = np.nonzero(keep)
    assert np.array_equal(row.cpu().numpy(), r2) and np.array_equal(col.cpu().numpy(), c2)
    assert np.array_equal(h.cpu().numpy(), ho[keep]) and np.array_equal(onv.cpu().numpy(), co[keep])
    assert np.array_equal(counts.cpu().numpy(), keep.sum(1))
    # ---- SAMPLE_SPACE in one kernel: table = the walkers, 20 000 of walker 0's kets, unrelated determinants ------
    g = np.random.default_rng(sorb)
    pick = g.choice(co.shape[1], 20000, replace=False)
    keys = np.unique(np.concatenate([onv_np, co[0, pick], O.pm01_to_onv(rand_occ(3000, sorb, no, no, seed=5), sorb)]), axis=0)
    wf = g.standard_normal(keys.shape[0]) + 1j * g.standard_normal(keys.shape[0])
    lut = pf.WavefunctionLUT(G(keys), G(wf), sorb, device=torch.device("cuda"))
    e, _, p0, _ = energy.local_energy(x, h1e, h2e, None, None, sorb, 2 * no, no, no, WF_LUT=lut, use_sample_space=True, dtype=torch.complex128)
    e_ref, p_ref = O.eloc_sample_space(onv_np, h1, h2, sorb, 2 * no, no, no, lut.bra_key.cpu().numpy(), lut.wf_value.cpu().numpy())
    np.testing.assert_array_equal(p0.cpu().numpy(), p_ref)
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=0, atol=TOL, err_msg=f"|E_loc|max = {float(np.abs(e_ref).max()):.6g} Ha")
    del comb, hm
    # ---- SIMPLE with the RBM on chip: alpha = 2 hidden units, the windowed kernel at these sizes ----------------------
    W, hb, vb = 0.01 * (g.random((H, sorb)) - 0.5), 0.01 * (g.random(H) - 0.5), 0.1 * (g.random(sorb) - 0.5)
    e_ref, p_ref = _rbm_eloc_logdomain(O, co[0], ho[0], sorb, W, hb, vb)
    e, p = cx.eloc_rbm(x[:1].contiguous(), h1e, h2e, cx.RBMTable(G(W), G(hb), G(vb)), sorb, 2 * no, no, no)
    np.testing.assert_allclose(p.cpu().numpy(), [p_ref], rtol=1e-10)
    np.testing.assert_allclose(e.cpu().numpy(), [e_ref], rtol=0, atol=TOL, err_msg=f"|E_loc| = {abs(e_ref):.6g} Ha")

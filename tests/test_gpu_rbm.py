"""Fused SIMPLE local energy with the on-chip real-RBM amplitude ratio (pynqs_eloc_rbm) against
 (i) the reference's own Python output on Fe2S2 (tests/golden/eloc_e2e_fe2s2.npz, eloc_simple / psi_simple) and
 (ii) the CPU oracle's materialise-and-forward restatement (oracle.eloc_simple_rbm) on seeded random problems with
      1, 2 and 3 ONV words, unequal alpha/beta counts and hidden units of both signs of theta.
Tolerance: 1e-8 Ha per determinant, scaled by the magnitude of the row's terms for the synthetic dense integrals."""
import numpy as np
import pytest
import torch

from conftest import golden, rand_occ, synth_integrals

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.fixture(scope="module")
def cx():
    from pynqs_amd import C_extension

    assert torch.cuda.is_available()
    return C_extension


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_fe2s2_matches_reference_python(cx, fe2s2):
    d = golden("eloc_e2e_fe2s2.npz")
    tab = cx.RBMTable(_dev(d["W"]), _dev(d["hb"]), _dev(d["vb"]))
    h1e, h2e = _dev(fe2s2["h1e"]), _dev(fe2s2["h2e"])
    eloc, psi = cx.eloc_rbm(_dev(d["x"]), h1e, h2e, tab, 40, 30, 15, 15)
    np.testing.assert_allclose(eloc.cpu().numpy(), d["eloc_simple"], rtol=0, atol=TOL)
    np.testing.assert_allclose(psi.cpu().numpy(), d["psi_simple"], rtol=1e-12)
    # few walkers: a walker's tiles are cut over several workgroups (atomics path); one walker; no psi
    for n in (1, 3):
        e, p = cx.eloc_rbm(_dev(d["x"][:n]), h1e, h2e, tab, 40, 30, 15, 15, want_psi=False)
        assert p is None
        np.testing.assert_allclose(e.cpu().numpy(), d["eloc_simple"][:n], rtol=0, atol=TOL)
    # CPU tensors are staged through the GPU
    e, p = cx.eloc_rbm(torch.from_numpy(d["x"][:4]), h1e, h2e, tab, 40, 30, 15, 15)
    assert e.device.type == "cpu"
    np.testing.assert_allclose(e.numpy(), d["eloc_simple"][:4], rtol=0, atol=TOL)


@pytest.mark.parametrize("sorb,noA,noB,H,n", [
    (8, 2, 2, 16, 36), (12, 3, 2, 24, 40), (12, 2, 4, 7, 33), (16, 5, 3, 32, 50), (10, 1, 1, 20, 25), (10, 4, 4, 5, 25),
    (2, 1, 1, 3, 1), (4, 1, 0, 6, 2), (64, 4, 3, 96, 9), (66, 3, 4, 70, 7), (128, 2, 3, 60, 5), (130, 3, 2, 64, 4),
])
def test_random_against_oracle(cx, sorb, noA, noB, H, n):
    from oracle import oracle

    h1, h2 = synth_integrals(sorb)
    occ = rand_occ(n, sorb, noA, noB, seed=sorb * 100 + noA)
    bra_cpu = oracle.pm01_to_onv(occ, sorb)
    g = np.random.default_rng(sorb + H)
    W = 0.3 * (g.random((H, sorb)) - 0.5)
    hb = 4.0 * (g.random(H) - 0.5)  # thetas of both signs, some large
    vb = 0.2 * (g.random(sorb) - 0.5)
    e_ref, p_ref = oracle.eloc_simple_rbm(bra_cpu, h1, h2, sorb, noA + noB, noA, noB, W, hb, vb)
    tab = cx.RBMTable(_dev(W), _dev(hb), _dev(vb))
    e, p = cx.eloc_rbm(_dev(bra_cpu), _dev(h1), _dev(h2), tab, sorb, noA + noB, noA, noB)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref, rtol=1e-11)
    scale = max(1.0, float(np.abs(e_ref).max()))
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=0, atol=TOL * scale)


def test_no_visible_bias_and_large_theta(cx):
    """visible_bias = None is a zero bias; |theta| ~ 40 must not overflow (cosh(40)^H would)."""
    from oracle import oracle

    sorb, noA, noB, H, n = 12, 3, 3, 20, 16
    h1, h2 = synth_integrals(sorb)
    bra_cpu = oracle.pm01_to_onv(rand_occ(n, sorb, noA, noB, seed=5), sorb)
    g = np.random.default_rng(11)
    W = 0.2 * (g.random((H, sorb)) - 0.5)
    hb = np.where(np.arange(H) % 2 == 0, 30.0, -30.0) + g.random(H)
    e_ref, _ = oracle.eloc_simple_rbm(bra_cpu, h1, h2, sorb, 6, noA, noB, W, hb, np.zeros(sorb))
    tab = cx.RBMTable(_dev(W), _dev(hb), None)
    e, p = cx.eloc_rbm(_dev(bra_cpu), _dev(h1), _dev(h2), tab, sorb, 6, noA, noB)
    assert torch.isfinite(e).all() and torch.isfinite(p).all()
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=0, atol=TOL * max(1.0, float(np.abs(e_ref).max())))


def test_empty_and_errors(cx, fe2s2):
    d = golden("eloc_e2e_fe2s2.npz")
    tab = cx.RBMTable(_dev(d["W"]), _dev(d["hb"]), _dev(d["vb"]))
    h1e, h2e = _dev(fe2s2["h1e"]), _dev(fe2s2["h2e"])
    e, p = cx.eloc_rbm(torch.empty((0, 8), dtype=torch.uint8, device="cuda"), h1e, h2e, tab, 40, 30, 15, 15)
    assert e.shape == (0,) and p.shape == (0,)
    with pytest.raises(RuntimeError):
        cx.eloc_rbm(_dev(d["x"]), h1e.float(), h2e.float(), tab, 40, 30, 15, 15)
    with pytest.raises(RuntimeError):
        cx.RBMTable(_dev(d["W"]).float(), _dev(d["hb"]).float(), None)
    with pytest.raises(RuntimeError):
        cx.RBMTable(_dev(d["W"]), _dev(d["hb"][:-1]), None)


@pytest.mark.parametrize("sorb,noA,noB,H,n", [(128, 3, 2, 400, 5), (184, 2, 2, 368, 3), (66, 5, 5, 700, 4), (40, 15, 15, 1200, 2)])
def test_windowed_kernel_against_oracle(cx, sorb, noA, noB, H, n):
    """sorb x num_hidden beyond the LDS: the windowed kernel (hidden units streamed through LDS in windows, one tile
    per wave and round) against the oracle's materialise-and-forward result."""
    from oracle import oracle
    from pynqs_amd import _native as N

    h1, h2 = synth_integrals(sorb)
    occ = rand_occ(n, sorb, noA, noB, seed=sorb + H)
    bra_cpu = oracle.pm01_to_onv(occ, sorb)
    g = np.random.default_rng(sorb * 7 + H)
    W = 0.05 * (g.random((H, sorb)) - 0.5)
    hb = 2.0 * (g.random(H) - 0.5)
    vb = 0.2 * (g.random(sorb) - 0.5)
    assert N.lib().pynqs_eloc_rbm_supported(sorb, noA + noB, noA, noB, H) == 1
    # reference in the log domain: with > 1000 hidden units psi = prod 2cosh(theta) itself overflows float64 (the
    # oracle, like the reference's rbm.py, then returns inf / nan) while the ratios stay finite
    comb, hm = oracle.comb_hij_fused(bra_cpu, h1, h2, sorb, noA + noB, noA, noB)
    xs = oracle.onv_to_pm1(comb.reshape(-1, comb.shape[-1]), sorb)
    th = xs @ W.T + hb
    lnpsi = (xs @ vb + (np.abs(th) + np.log1p(np.exp(-2.0 * np.abs(th)))).sum(1)).reshape(n, -1)
    e_ref = (hm * np.exp(lnpsi - lnpsi[:, :1])).sum(1)
    tab = cx.RBMTable(_dev(W), _dev(hb), _dev(vb))
    e, p = cx.eloc_rbm(_dev(bra_cpu), _dev(h1), _dev(h2), tab, sorb, noA + noB, noA, noB)
    with np.errstate(over="ignore"):
        p_ref = np.exp(lnpsi[:, 0])
    if np.isfinite(p_ref).all():
        np.testing.assert_allclose(p.cpu().numpy(), p_ref, rtol=1e-10)
        e_orc, _ = oracle.eloc_simple_rbm(bra_cpu, h1, h2, sorb, noA + noB, noA, noB, W, hb, vb)
        np.testing.assert_allclose(e_orc, e_ref, rtol=0, atol=TOL * max(1.0, float(np.abs(e_ref).max())))
    else:
        assert torch.isinf(p).all()
    scale = max(1.0, float(np.abs(e_ref).max()))
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=0, atol=TOL * scale)


def test_end_to_end_vmc_lowers_the_energy():
    """examples/vmc_rbm_exact_sampling.py: fused E_loc -> statistics kernel -> gradient -> Adam on a sorb-8 problem
    with exact sampling: the variational energy must go down and stay above the exact ground state."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vmc_example", os.path.join(root, "examples", "vmc_rbm_exact_sampling.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        hist, e0 = mod.run(steps=80, log=lambda *_: None)
    finally:
        torch.set_default_dtype(torch.float32)
    assert hist[-1] < hist[0] - 0.05
    assert min(hist) >= e0 - 1e-9
    assert hist[-1] - e0 < 0.6 * (hist[0] - e0)

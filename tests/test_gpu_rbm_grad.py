"""pynqs_rbm_grad / grad.FusedRbmGrad: the analytic energy-gradient estimator of the RBM amplitudes (vmc/grad/energy_grad.py:118-184 on
vmc/ansatz/rbm/rbm.py:186-211) against the same estimator through autograd (pynqs_amd.grad.grad, itself pinned on vectors captured from
the reference: tests/test_gpu_grad.py).  Tolerance: 1e-11 relative to the largest gradient entry; the loss to 1e-10."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _walkers(n, sorb, no, seed):
    import bench as B

    return B.synth_walkers(n, sorb, no, no, seed)


def _modules(kind, sorb, H, dev, seed):
    from pynqs_amd.rbm import ComplexRBM, RealRBM

    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.rand(*s, generator=g, dtype=torch.float64) - 0.5)  # noqa: E731
    if kind == "complex":
        return ComplexRBM(0.3 * r(H, sorb, 2), 0.4 * r(H, 2), 0.2 * r(sorb, 2)).to(dev)
    return RealRBM(0.3 * r(H, sorb), 0.4 * r(H), 0.2 * r(sorb)).to(dev)


@pytest.mark.parametrize("kind,sorb,no,H,n,eloc_cplx,use_pow", [
    ("complex", 40, 15, 40, 1000, True, False), ("complex", 40, 15, 37, 777, True, True), ("complex", 12, 3, 5, 64, True, False),
    ("complex", 120, 30, 70, 300, True, False), ("complex", 184, 46, 33, 130, True, False),
    ("real", 40, 15, 80, 1000, False, False), ("real", 40, 15, 80, 500, True, True), ("real", 72, 6, 9, 65, False, False)])
def test_fused_gradient_matches_autograd(kind, sorb, no, H, n, eloc_cplx, use_pow):
    from pynqs_amd import C_extension as cx, grad as G

    dev = torch.device("cuda")
    m = _modules(kind, sorb, H, dev, 3)
    x = _walkers(n, sorb, no, 17).to(dev)
    g = torch.Generator().manual_seed(5)
    prob = torch.rand(n, generator=g, dtype=torch.float64); prob = (prob / prob.sum()).to(dev)
    eloc = (torch.randn(n, generator=g, dtype=torch.float64) - 100.0)
    if eloc_cplx:
        eloc = torch.complex(eloc, 0.1 * torch.randn(n, generator=g, dtype=torch.float64))
    eloc = eloc.to(dev)
    e_tot = (prob * eloc).sum()
    pw = (0.5 + torch.rand(n, generator=g, dtype=torch.float64)).to(dev) if use_pow else 1.0
    dtype = torch.complex128 if (kind == "complex" or eloc_cplx) else torch.float64
    states = cx.onv_to_tensor(x, sorb).to(torch.float64)
    for p in m.parameters():
        p.grad = None
    loss_ref = G.grad(m, states, prob, eloc, e_tot, pw if not use_pow else pw.to(dtype), dtype)
    want = [p.grad.clone() for p in m.parameters()]
    fg = G.FusedRbmGrad(m, sorb)
    loss = fg(x, prob, eloc, e_tot, pw)
    scale = max(float(w.abs().max()) for w in want)
    for p, w in zip(m.parameters(), want):
        assert p.grad.shape == w.shape
        np.testing.assert_allclose(p.grad.cpu().numpy(), w.cpu().numpy(), rtol=0, atol=1e-11 * scale)
    np.testing.assert_allclose(float(loss), float(loss_ref), rtol=0, atol=1e-10 * max(1.0, abs(float(loss_ref))))
    first = [p.grad.clone() for p in m.parameters()]
    fg(x, prob, eloc, e_tot, pw)
    assert all(torch.equal(a, p.grad) for a, p in zip(first, m.parameters()))  # fixed order of additions


def test_fused_gradient_refuses_other_modules_and_handles_no_walkers():
    from pynqs_amd import grad as G
    from pynqs_amd.rbm import RealRBM

    dev = torch.device("cuda")
    z = torch.zeros
    with pytest.raises(ValueError):
        G.FusedRbmGrad(RealRBM(z(4, 8, dtype=torch.float64), z(4, dtype=torch.float64), z(8, dtype=torch.float64), rbm_type="tanh").to(dev), 8)
    with pytest.raises(ValueError):
        G.FusedRbmGrad(torch.nn.Linear(8, 1).double().to(dev), 8)
    m = _modules("complex", 8, 4, dev, 1)
    fg = G.FusedRbmGrad(m, 8)
    loss = fg(torch.zeros((0, 8), dtype=torch.uint8, device=dev), torch.zeros(0, dtype=torch.float64, device=dev),
              torch.zeros(0, dtype=torch.complex128, device=dev), torch.zeros((), dtype=torch.complex128, device=dev))
    assert float(loss) == 0.0 and all(float(p.grad.abs().max()) == 0.0 for p in m.parameters())

"""pynqs_amd.grad.grad against parameter gradients captured from the reference's vmc/grad/energy_grad.py:118-184 (run under
DistributedDataParallel on its own RBM / on the complex128 module; tests/golden/make_golden_r2.py -> grad_fe2s2.npz).
CPU: the estimator is host logic (autograd).  The world_size-2 form is in test_distributed_cpu.py, the CUDA form below (gpu)."""
import numpy as np
import pytest
import torch

from conftest import GRAD_CASES, grad_case

RTOL = 1e-10


def _run(device):
    from pynqs_amd.grad import grad

    for kind, amd, use_pow in GRAD_CASES:
        m, states, prob, eloc, e_total, powr, dt, amd_, want = grad_case(kind, amd, use_pow, device)
        loss = grad(m, states, prob, eloc, e_total, powr, dt, amd_)
        assert loss.shape == (1,) and loss.dtype == torch.float64
        got = dict(m.named_parameters())
        assert set(got) == set(want)
        for name, w in want.items():
            scale = np.abs(w).max()
            np.testing.assert_allclose(got[name].grad.cpu().numpy(), w, rtol=RTOL, atol=RTOL * scale, err_msg=f"{kind} {amd} {name}")


def test_grad_matches_reference_cpu():
    _run("cpu")


@pytest.mark.gpu
def test_grad_matches_reference_gpu():
    _run("cuda")


@pytest.mark.gpu
def test_graphed_grad_matches_reference_gpu():
    """GraphedGrad (forward + backward replayed from a HIP graph) against the same reference gradients; a second call with other
    inputs replays the same graph."""
    from pynqs_amd.grad import GraphedGrad

    for kind, amd, use_pow in GRAD_CASES:
        m, states, prob, eloc, e_total, powr, dt, _amd, want = grad_case(kind, amd, use_pow, "cuda")
        gg = GraphedGrad(m, states.size(0), states.size(1), dt, use_pow=bool(use_pow))
        gg(states, prob * 0.5, eloc + 1.0, e_total, powr)  # other inputs first: the graph must not bake values in
        loss = gg(states, prob, eloc, e_total, powr)
        assert loss.shape == (1,)
        for name, p in m.named_parameters():
            w = want[name]
            np.testing.assert_allclose(p.grad.cpu().numpy(), w, rtol=RTOL, atol=RTOL * np.abs(w).max(), err_msg=f"{kind} {name}")


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_fused_rbm_grad_matches_reference_gpu(world):
    """pynqs_rbm_grad / grad.FusedRbmGrad (the analytic estimator the default bench step runs) DIRECTLY against the gradients the reference's
    energy_grad.py:118-184 produced under DDP (grad_fe2s2.npz), all four cases, at world size 1 and on the two ranks' shards of the world-size-2
    run (DDP averages the ranks' gradients: energy_grad.py:167-181)."""
    from conftest import golden
    from pynqs_amd.grad import FusedRbmGrad

    onv = golden("eloc_e2e_fe2s2.npz")["x"]
    for kind, amd, use_pow in GRAD_CASES:
        acc, want = None, None
        for rank in range(world):
            m, states, prob, eloc, e_total, powr, dt, _amd, want = grad_case(kind, amd, use_pow, "cuda", rank, world)
            k, res = divmod(32, world)
            b = rank * k + min(rank, res)
            x = torch.from_numpy(np.ascontiguousarray(onv[b:b + states.size(0)])).cuda()
            fg = FusedRbmGrad(m, 40)
            loss = fg(x, prob, eloc, e_total, powr)
            assert loss.shape == (1,) and loss.dtype == torch.float64
            got = {name: p.grad.detach().cpu().numpy().copy() for name, p in m.named_parameters()}
            acc = got if acc is None else {nm: acc[nm] + got[nm] for nm in got}
        assert set(acc) == set(want)
        for name, w in want.items():
            np.testing.assert_allclose(acc[name] / world, w, rtol=RTOL, atol=RTOL * np.abs(w).max(), err_msg=f"{kind} {amd} ws{world} {name}")

"""GFMC step and sampler natives on the GPU against vectors captured from the reference's Python (gfmc/walker.py:167-279,
cpp_src/tensor/cpu_tensor.cpp:537-556; tests/golden/make_golden_r2.py -> gfmc_fe2s2.npz, sampler_merge.npz)."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_green", [True, False])
@pytest.mark.parametrize("fused_sample", [True, False])
@pytest.mark.parametrize("tag", ["a", "b"])
def test_green_kernel_and_move_match_reference_python(fe2s2, tag, fused_sample, fused_green):
    """_calculate_green_kernel (fixed-node effective Hamiltonian, sign-flip potential, clamp of negative diagonal kernels:
    case b clamps 5 of the 16 walkers) and sample_update with the reference's uniforms.  fused_green: the whole row from
    pynqs_green_rbm (the trial function is a real RBM) and the move from the column's rank, nothing materialised; else comb + module."""
    from pynqs_amd import gfmc, public_function as pf
    from pynqs_amd.rbm import RealRBM

    d, e0 = golden("gfmc_fe2s2.npz"), golden("eloc_e2e_fe2s2.npz")
    dev = torch.device("cuda")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    old, old_g = gfmc.FUSED_SAMPLE, gfmc.FUSED_GREEN
    try:
        gfmc.FUSED_SAMPLE, gfmc.FUSED_GREEN = fused_sample, fused_green
        rbm = RealRBM(T(e0["W"]), T(e0["hb"]), T(e0["vb"])).to(dev)
        ab = lambda x, func: pf.ansatz_batch(func, x, 100000, 40, dev, torch.double)  # noqa: E731
        x = T(d["x"])
        eloc, gk, comb, stop, mask = gfmc.green_kernel(x, float(d[tag + "_Lambda"]), T(fe2s2["h1e"]), T(fe2s2["h2e"]), rbm, ab, 40, 30, 15, 15,
                                                       torch.double, None, True)
        assert stop is False
        assert isinstance(comb, gfmc.CombRows) == fused_green
        if fused_green and tag == "a" and fused_sample:
            from pynqs_amd import C_extension as cx

            assert np.array_equal(comb.materialize().cpu().numpy(), cx.get_comb_hij_fused(x, T(fe2s2["h1e"]), T(fe2s2["h2e"]), 40, 30, 15, 15)[0].cpu().numpy())
        np.testing.assert_allclose(eloc.cpu().numpy(), d[tag + "_eloc"], rtol=0, atol=1e-8)
        assert np.array_equal(mask.cpu().numpy(), d[tag + "_mask"])
        np.testing.assert_allclose(gk[:4].cpu().numpy(), d[tag + "_gk4"], rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(gk[:, 0].cpu().numpy(), d[tag + "_gk_col0"], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(gk.sum(-1).cpu().numpy(), d[tag + "_gk_rowsum"], rtol=1e-11)
        x_new, w_new, beta, acc = gfmc.sample_update(x, T(d[tag + "_weight"]), comb, gk, T(d[tag + "_rand"]))
        assert np.array_equal(x_new.cpu().numpy(), d[tag + "_x_new"])
        np.testing.assert_allclose(w_new.cpu().numpy(), d[tag + "_w_new"], rtol=1e-11)
        np.testing.assert_allclose(beta.cpu().numpy().reshape(-1), d[tag + "_beta"].reshape(-1), rtol=1e-11)
        assert acc == int(d[tag + "_accept"])
    finally:
        gfmc.FUSED_SAMPLE, gfmc.FUSED_GREEN = old, old_g
        torch.set_default_dtype(old_dt)


def test_merge_rank_sample_matches_reference():
    from pynqs_amd import C_extension as cx

    sm = golden("sampler_merge.npz")
    dev = torch.device("cuda")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    got = cx.merge_rank_sample(T(sm["mrs_inv"]), T(sm["mrs_counts"]), T(sm["mrs_split"]), int(sm["mrs_length"]))
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), sm["mrs_out"])
    # the merge of the sampler protocol itself: tensor_to_onv of the occupations, unique + merged counts as rank 0 of the reference
    for t in (0, 1):
        occ = np.concatenate([sm[f"tree{t}_r{r}_occ"] for r in (0, 1)])
        cnt = np.concatenate([sm[f"tree{t}_r{r}_counts"] for r in (0, 1)])
        onv = cx.tensor_to_onv(T(occ), 40)
        if t == 0:
            from pynqs_amd.sample_comm import torch_unique_index

            mu, inv, idx, _ = torch_unique_index(onv)
            mc = cx.merge_rank_sample(inv.contiguous(), T(cnt), T(np.array([0, sm["tree0_r0_occ"].shape[0], occ.shape[0]])), mu.size(0))
        else:
            mu, mc = onv, T(cnt)
        assert np.array_equal(mc.cpu().numpy(), sm[f"tree{t}_r0_all_counts"])
        assert np.array_equal(mu.cpu().numpy(), np.concatenate([sm[f"tree{t}_r{r}_unique_rank"] for r in (0, 1)]))


@pytest.mark.parametrize("sorb,noA,noB,H,n,kind", [(12, 3, 2, 24, 40, "real"), (16, 4, 4, 20, 30, "tanh"), (66, 3, 4, 70, 7, "real"),
                                                   (130, 3, 2, 64, 4, "tanh"), (128, 3, 2, 400, 5, "real")])
def test_fused_green_row_random_systems(sorb, noA, noB, H, n, kind):
    """pynqs_green_rbm + pynqs_gfmc_sample_rank (1-3 ONV words, unequal alpha / beta, resident and windowed kernels, amplitudes of
    both signs for "tanh") against the materialising path of the same module -- which test_green_kernel_and_move_match_reference_python
    pins to the reference on Fe2S2: row, E_loc, clamp mask, and the move for the same uniforms."""
    from conftest import rand_occ, synth_integrals
    from oracle import oracle
    from pynqs_amd import gfmc, public_function as pf
    from pynqs_amd.rbm import RealRBM

    dev = torch.device("cuda")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    h1, h2 = synth_integrals(sorb)
    bra = oracle.pm01_to_onv(rand_occ(n, sorb, noA, noB, seed=3 * sorb + H), sorb)
    x = T(bra.view(np.uint8).reshape(n, -1))
    g = np.random.default_rng(sorb * 13 + H)
    W, hb, vb = 0.05 * (g.random((H, sorb)) - 0.5), 1.0 * (g.random(H) - 0.5), 0.3 * (g.random(sorb) - 0.5)
    old_dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    old_g = gfmc.FUSED_GREEN
    try:
        m = RealRBM(T(W), T(hb), T(vb), rbm_type=kind).to(dev)
        ab = lambda xx, func: pf.ansatz_batch(func, xx, 100000, sorb, dev, torch.double)  # noqa: E731
        out = {}
        rnd = T(g.random((n, 1)))
        wgt = T(g.random(n) + 0.5)
        for fused in (True, False):
            gfmc.FUSED_GREEN = fused
            # Lambda between the rows' diagonal kernels: some walkers clamp
            eloc, gk, comb, _, neg = gfmc.green_kernel(x, 0.0, T(h1), T(h2), m, ab, sorb, noA + noB, noA, noB, torch.double, None, True)
            x_new, w_new, beta, acc = gfmc.sample_update(x, wgt, comb, gk, rnd)
            out[fused] = [t.cpu().numpy() for t in (eloc, gk, neg, x_new, w_new, beta)] + [acc]
        scale = max(1.0, float(np.abs(out[False][1]).sum(1).max()))
        np.testing.assert_allclose(out[True][0], out[False][0], rtol=0, atol=1e-8 * scale)
        np.testing.assert_allclose(out[True][1], out[False][1], rtol=1e-9, atol=1e-11 * scale)
        assert np.array_equal(out[True][2], out[False][2])
        assert np.array_equal(out[True][3], out[False][3])
        np.testing.assert_allclose(out[True][4], out[False][4], rtol=1e-10)
        assert out[True][6] == out[False][6]
    finally:
        gfmc.FUSED_GREEN = old_g
        torch.set_default_dtype(old_dt)


def test_fixed_node_gfmc_converges_to_the_fixed_node_energy():
    """examples/gfmc_rbm_fixed_node.py: walkers from |psi_T|^2, 100 generations of (fused Green's-function row, rank move, resampling);
    the mixed estimator must land on the lowest eigenvalue of the fixed-node Hamiltonian diagonalised in the full determinant space
    (statistical error ~0.015 with 8192 walkers), between the exact ground state and the trial function's variational energy."""
    import importlib.util
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "gfmc_rbm_fixed_node.py")
    spec = importlib.util.spec_from_file_location("gfmc_rbm_fixed_node", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = torch.get_default_dtype()
    try:
        e_exact, e_fn, e_gfmc, e_vmc = mod.run(generations=100, walkers=8192, burn_in=30, log=lambda *a: None)
    finally:
        torch.set_default_dtype(old)
    assert e_exact <= e_fn + 1e-9 and e_fn <= e_vmc + 1e-9
    assert abs(e_gfmc - e_fn) < 0.08, (e_gfmc, e_fn)
    assert e_gfmc < e_vmc - 0.5

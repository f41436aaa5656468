#!/usr/bin/env python3
"""Third batch of golden vectors, captured from the REFERENCE's own Python running on its own CPU extension (development container
only; see make_golden.py / make_golden_r2.py for how the reference is built and imported).  Only DATA is written.

  eloc_reduce_sampled_fe2s2.npz  the semi-stochastic REDUCE local energies (vmc/energy/eloc.py:257-296, flip.py:205-236: eps = 1e-2,
        eps_sample = 200) for EVERY form -- plain, + look-up table, complex128 module, spin-flip projected, multi-psi, both, <S-S+> --
        with torch.multinomial answering with the draws of tests/golden/reduce_draws_fe2s2.npz (what pynqs_amd's kernel draws for
        torch.manual_seed(20240); written on the GPU box by tests/golden/dump_reduce_draws.py), so that the GPU tests can compare the
        whole path at 1e-8 Ha instead of statistically.
  eloc_bdg_rnn_fe2s2.npz         local energies with the amplitude the Fe2S2 example itself optimises: the reference's Graph_MPS_RNN
        (vmc/ansatz/rnn/graph_mpsrnn.py:239, dcut 20, complex128) with the shipped parameters example/Fe2S2/fe2s2-OO-dcut-20-focus-1e-8.pth
        on the 32 determinants of ci_space it weighs most: psi on every x' the REDUCE (eps = 1e-2) and semi-stochastic selections touch (|psi| spans > 8 decades),
        the reference's REDUCE, semi-stochastic REDUCE and SAMPLE_SPACE local energies.  The GPU tests feed psi through a table-backed module.

usage: python tests/golden/make_golden_r3.py --stage walkers            (here)   -> bdg_rnn_walkers_fe2s2.npz
       gpurun -- python tests/golden/dump_reduce_draws.py              (GPU box) -> reduce_draws_fe2s2.npz, reduce_draws_bdg_rnn_fe2s2.npz
       python tests/golden/make_golden_r3.py                           (here)   -> the fixtures
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import make_golden_r2 as R2  # noqa: E402


class FixedDraws:
    """torch.multinomial replaced by the draws of the fixture: row i of the answer lists column c `hits` times."""

    def __init__(self, d, n_walkers):
        N = int(d["eps_sample"])
        self.rows = torch.zeros((n_walkers, N), dtype=torch.int64)
        fill = [0] * n_walkers
        for w, c, h in zip(d["draw_walker"].tolist(), d["draw_col"].tolist(), d["draw_hits"].tolist()):
            self.rows[w, fill[w]:fill[w] + h] = c
            fill[w] += h
        assert all(f == N for f in fill)
        self.calls = 0

    def __call__(self, prob, num_samples, replacement=False, **kw):
        assert replacement and num_samples == self.rows.size(1) and prob.size(0) == self.rows.size(0)
        # every drawn column must be drawable (sub-eps, non-zero probability)
        assert bool((prob.gather(1, self.rows) > 0).all()), "a fixture draw has zero probability in the reference"
        self.calls += 1
        return self.rows.clone()


def section_sampled(I, out_dir):
    from utils.public_function import SpinProjection, WavefunctionLUT, ansatz_batch
    from utils.pyscf_helper.operator import spin_raising
    from vmc.ansatz.rbm.rbm import RBMWavefunction
    from vmc.energy.eloc import local_energy

    d = np.load(f"{HERE}/reduce_draws_fe2s2.npz")
    sorb, nele, noA, noB = I["sorb"], I["nele"], I["noA"], I["noB"]
    h1, h2, x, ci = I["h1e"], I["h2e"], I["x"], I["ci"]
    assert np.array_equal(d["x"], x.numpy())
    eps, N = float(d["eps"]), int(d["eps_sample"])
    (W2, hb2, vb2), (Wc, hbc, vbc) = R2.second_rbm_params(sorb)
    rbm = RBMWavefunction(sorb, alpha=2, rbm_type="real"); rbm.init(I["hb"].clone(), I["W"].clone(), I["vb"].clone())
    extra = RBMWavefunction(sorb, alpha=1, rbm_type="real"); extra.init(hb2.clone(), W2.clone(), vb2.clone())
    crbm = R2.complex_module(Wc, hbc, vbc)
    multi = R2.Holder(rbm, extra)
    SpinProjection.init(nele, 0)
    extra_norm = torch.tensor(1.3, dtype=torch.float64)
    extra_norm_m = torch.tensor(1.3 * 2.0**40, dtype=torch.float64)
    h1s, h2s = spin_raising(sorb, c1=1.0)
    draws = FixedDraws(d, x.size(0))
    real_multinomial = torch.multinomial
    torch.multinomial = draws

    def ab(dt):
        return lambda x_, func: ansatz_batch(func, x_, 100000, sorb, torch.device("cpu"), dt)

    def le(ansatz, dt=torch.double, **kw):
        e, s, p, _ = local_energy(x, h1, h2, ansatz, ab(dt), sorb, nele, noA, noB, dtype=dt, reduce_psi=True, eps=eps, eps_sample=N, **kw)
        return e.detach().numpy(), s.detach().numpy(), p.detach().numpy()

    keys = ci[:700].contiguous()
    with torch.no_grad():
        wf = ab(torch.double)(keys, rbm)
    lut = WavefunctionLUT(keys, wf, sorb, device="cpu")
    out = dict(x=x.numpy(), eps=eps, eps_sample=N, torch_seed=int(d["torch_seed"]), eta=SpinProjection.eta, extra_norm=float(extra_norm),
               extra_norm_multi=float(extra_norm_m), lut_keys=keys.numpy(), lut_wf=wf.numpy(), h1e_spin=h1s.numpy(), h2e_spin=h2s.numpy())
    try:
        for name, args in (("plain", (rbm, torch.double, {})), ("lut", (rbm, torch.double, dict(WF_LUT=lut))), ("complex", (crbm, torch.complex128, {})),
                           ("flip", (rbm, torch.double, dict(use_spin_flip=True, extra_norm=extra_norm))),
                           ("flip_lut", (rbm, torch.double, dict(use_spin_flip=True, extra_norm=extra_norm, WF_LUT=lut))),
                           ("multi", (multi, torch.double, dict(use_multi_psi=True, extra_norm=extra_norm_m))),
                           ("flip_multi", (multi, torch.double, dict(use_spin_flip=True, use_multi_psi=True, extra_norm=extra_norm_m))),
                           ("spin_raising", (rbm, torch.double, dict(use_spin_raising=True, h1e_spin=h1s, h2e_spin=h2s))),
                           ("flip_spin_raising", (rbm, torch.double, dict(use_spin_flip=True, extra_norm=extra_norm, use_spin_raising=True, h1e_spin=h1s, h2e_spin=h2s)))):
            ansatz, dt, kw = args
            out["eloc_" + name], out["sloc_" + name], out["psi_" + name] = le(ansatz, dt, **kw)
    finally:
        torch.multinomial = real_multinomial
    assert draws.calls == 9
    np.savez_compressed(f"{out_dir}/eloc_reduce_sampled_fe2s2.npz", **out)
    print("sampled REDUCE:", {k: float(np.abs(v).max()) for k, v in out.items() if k.startswith("eloc_")})
    return d


def build_rnn(I):
    import networkx as nx
    from vmc.ansatz.rnn.graph_mpsrnn import Graph_MPS_RNN

    cwd = os.getcwd()
    os.chdir(MG.REF)
    try:
        g = nx.read_graphml("./example/Fe2S2/Fe2S2-maxdes-0.graphml")
        return Graph_MPS_RNN(use_symmetry=True, param_dtype=torch.complex128, hilbert_local=4, nqubits=I["sorb"], nele=I["nele"], device="cpu", dcut=20,
                             graph=g, params_file="./example/Fe2S2/fe2s2-OO-dcut-20-focus-1e-8.pth")
    finally:
        os.chdir(cwd)


def stage_walkers(I, out_dir):
    """The 32 determinants of the file's ci_space on which the shipped BDG-RNN parameters put the most weight: the walkers a VMC run with
    this amplitude would actually hold (on ci_space[:32], where |psi| ~ 1e-20 next to neighbours of 1e-11, E_loc is ~1e12 Ha and no
    absolute tolerance means anything)."""
    from utils.public_function import ansatz_batch

    rnn = build_rnn(I)
    with torch.no_grad():
        psi = ansatz_batch(rnn, I["ci"], 4096, I["sorb"], torch.device("cpu"), torch.complex128)
    top = torch.argsort(psi.abs(), descending=True)[:32]
    np.savez_compressed(f"{out_dir}/bdg_rnn_walkers_fe2s2.npz", x=I["ci"][top].numpy(), psi=psi[top].numpy(), ci_index=top.numpy())
    print("BDG-RNN walkers: |psi| from", float(psi[top].abs().min()), "to", float(psi[top].abs().max()), "of all ci_space:", float(psi.abs().max()))


def section_bdg_rnn(I, out_dir):
    from utils.public_function import WavefunctionLUT, ansatz_batch
    from vmc.energy.eloc import local_energy

    sorb, nele, noA, noB = I["sorb"], I["nele"], I["noA"], I["noB"]
    h1, h2 = I["h1e"], I["h2e"]
    d = np.load(f"{HERE}/reduce_draws_bdg_rnn_fe2s2.npz")
    x = torch.from_numpy(np.load(f"{HERE}/bdg_rnn_walkers_fe2s2.npz")["x"])
    assert np.array_equal(d["x"], x.numpy())
    eps, N = float(d["eps"]), int(d["eps_sample"])
    rnn = build_rnn(I)
    seen = {}

    def recording(xpm):
        """the module, remembering every (determinant, psi) it is asked for"""
        with torch.no_grad():
            psi = rnn(xpm)
        bits = (xpm > 0).to(torch.uint8).numpy()
        packed = np.packbits(bits, axis=1, bitorder="little")
        keys = np.zeros((packed.shape[0], 8), dtype=np.uint8)
        keys[:, : packed.shape[1]] = packed
        for k, v in zip(keys.view(np.uint64).reshape(-1).tolist(), psi.numpy().tolist()):
            seen[k] = v
        return psi

    ab = lambda x_, func: ansatz_batch(func, x_, 100000, sorb, torch.device("cpu"), torch.complex128)  # noqa: E731

    def le(**kw):
        e, s, p, _ = local_energy(x, h1, h2, recording, ab, sorb, nele, noA, noB, dtype=torch.complex128, **kw)
        return e.detach().numpy(), p.detach().numpy()

    out = dict(x=x.numpy(), eps=eps, eps_sample=N, torch_seed=int(d["torch_seed"]))
    out["eloc_reduce"], out["psi_reduce"] = le(reduce_psi=True, eps=eps, eps_sample=0)
    draws = FixedDraws(d, x.size(0))
    real_multinomial = torch.multinomial
    torch.multinomial = draws
    try:
        out["eloc_sampled"], out["psi_sampled"] = le(reduce_psi=True, eps=eps, eps_sample=N)
    finally:
        torch.multinomial = real_multinomial
    # SAMPLE_SPACE over the determinants seen so far (a few thousand: the walkers and their important neighbours)
    keys = torch.from_numpy(np.array(sorted(seen), dtype=np.uint64).view(np.uint8).reshape(-1, 8).copy())
    vals = torch.tensor([seen[k] for k in sorted(seen)], dtype=torch.complex128)
    lut = WavefunctionLUT(keys, vals, sorb, device="cpu")
    out["eloc_ss"], out["psi_ss"] = le(use_sample_space=True, WF_LUT=lut, index=(0, x.size(0)))
    out["table_keys"], out["table_psi"] = keys.numpy(), vals.numpy()
    np.savez_compressed(f"{out_dir}/eloc_bdg_rnn_fe2s2.npz", **out)
    a = np.abs(vals.numpy())
    print(f"BDG-RNN: {len(seen)} determinants, |psi| from {a[a > 0].min():.3e} to {a.max():.3e}; max|E_loc| reduce {np.abs(out['eloc_reduce']).max():.6g}, "
          f"sampled {np.abs(out['eloc_sampled']).max():.6g}, sample space {np.abs(out['eloc_ss']).max():.6g}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/refbuild")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--stage", default="fixtures", choices=["walkers", "fixtures"],
                    help="walkers: choose the BDG-RNN walkers (before tests/golden/dump_reduce_draws.py runs on the GPU box); fixtures: everything else")
    a = ap.parse_args()
    R2.harness(a.scratch)
    torch.set_default_dtype(torch.double)
    I = R2.load_inputs()
    if a.stage == "walkers":
        stage_walkers(I, a.out)
    else:
        section_sampled(I, a.out)
        section_bdg_rnn(I, a.out)

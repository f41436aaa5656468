#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE itself.

Runs only in the development container (needs /root/reference); nothing here runs on the GPU
box.  The reference's CPU extension (cpp_src/{common,cpu,tensor}, CUDA/MAGMA glue dropped,
`GPU` undefined) is compiled in a scratch directory OUTSIDE this repository, once per
MAX_SORB_LEN in {1,2,3} (the single macro edit cpp_src/common/default.h:3 that README.md:42-44
tells users to make), and imported; its outputs on seeded inputs are stored as small .npz files.
For the end-to-end local-energy vectors the reference's Python package is imported from
/root/reference with two inert stand-ins for absent logging/typing packages (`loguru`,
`jaxtyping`) that carry no arithmetic.

Only DATA lands in tests/golden/: inputs and the reference's outputs.  No reference source.

usage: python tests/golden/make_golden.py [--scratch /tmp/refbuild]
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import importlib.util
import os
import re
import shutil
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def build_ref(scratch: str, L: int):
    D = os.path.join(scratch, f"L{L}")
    # distinct module names so that all three variants can live in one process
    name = "C_extension" if L == 1 else f"C_extension_L{L}"
    sos = glob.glob(os.path.join(D, name + ".so"))
    if not sos:
        if os.path.exists(D):
            shutil.rmtree(D)
        os.makedirs(D)
        for sub in ("common", "cpu", "tensor"):
            shutil.copytree(f"{REF}/cpp_src/{sub}", f"{D}/{sub}")
        os.remove(f"{D}/tensor/cuda_tensor.cpp")
        os.remove(f"{D}/tensor/interface_magma.cpp")
        p = f"{D}/common/default.h"
        s = open(p).read()
        open(p, "w").write(re.sub(r"#define MAX_SORB_LEN 1 ", f"#define MAX_SORB_LEN {L} ", s))
        import torch.utils.cpp_extension as E

        E.load(name=name, sources=sorted(glob.glob(f"{D}/*/*.cpp")), extra_include_paths=[D],
               extra_cflags=["-O3", "-fopenmp", "-std=c++17", "-UGPU"], extra_ldflags=["-fopenmp"],
               build_directory=D, verbose=False)
        sos = glob.glob(os.path.join(D, name + ".so"))
    spec = importlib.util.spec_from_file_location(name, sos[0])
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.MAX_SORB_LEN == L
    return m, sos[0]


def synth_integrals(sorb: int, seed: int = 1234):
    """dense uniform integrals straight in the packed layout."""
    g = torch.Generator().manual_seed(seed)
    h1 = torch.rand(sorb, sorb, generator=g, dtype=torch.float64) - 0.5
    h1 = (h1 + h1.T).reshape(-1)
    pair = sorb * (sorb - 1) // 2
    h2 = torch.rand(pair * (pair + 1) // 2, generator=g, dtype=torch.float64) - 0.5
    return h1, h2


def rand_occ(n, sorb, noA, noB, seed):
    g = np.random.default_rng(seed)
    occ = np.zeros((n, sorb), dtype=np.uint8)
    for i in range(n):
        occ[i, 2 * g.permutation(sorb // 2)[:noA]] = 1
        occ[i, 2 * g.permutation(sorb // 2)[:noB] + 1] = 1
    return occ


def all_dets(sorb, noA, noB):
    import itertools

    k = sorb // 2
    rows = []
    for ca in itertools.combinations(range(k), noA):
        for cb in itertools.combinations(range(k), noB):
            o = np.zeros(sorb, dtype=np.uint8)
            o[[2 * i for i in ca]] = 1
            o[[2 * i + 1 for i in cb]] = 1
            rows.append(o)
    return np.stack(rows)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_fe2s2():
    import numpy

    import numpy._core.multiarray as ncm

    # the file was pickled under numpy<2 (module path numpy.core.*): allow-list the array
    # reconstruction helpers under that legacy path instead of unpickling arbitrary code
    allow = [(ncm._reconstruct, "numpy.core.multiarray._reconstruct"), numpy.ndarray, numpy.dtype]
    allow += [type(numpy.dtype(t)) for t in ("float64", "uint8", "int64", "float32", "int32")]
    with torch.serialization.safe_globals(allow):
        e = torch.load(f"{REF}/example/Fe2S2/fe2s2-OO.pth", weights_only=True)
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/refbuild")
    args = ap.parse_args()
    os.makedirs(args.scratch, exist_ok=True)
    torch.set_default_dtype(torch.float64)
    mods = {L: build_ref(args.scratch, L) for L in (1, 2, 3)}
    m1 = mods[1][0]

    # ---- 0. docstring known answers (libs/C_extension.pyi:17-23,37-43,75-89,327-355) -----------
    bra = torch.tensor([[0b1100, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.uint8)
    onv, states = m1.get_comb_tensor(bra, 4, 2, 1, 1, True)
    assert onv[0, :, 0].tolist() == [12, 9, 6, 3]
    t2o = m1.tensor_to_onv(torch.tensor([1, 1, 1, 1, 0, 0, 0, 0], dtype=torch.uint8), 8)
    o2t = m1.onv_to_tensor(torch.tensor([[0b1111, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.uint8), 8)
    np.savez_compressed(f"{HERE}/docstring_examples.npz", comb_onv=onv.numpy(), comb_states=states.numpy(),
                        t2o=t2o.numpy(), o2t=o2t.numpy())

    # ---- 1. C1-sized exhaustive: sorb 8, 2a2b, all 36 determinants -----------------------------
    sorb, noA, noB = 8, 2, 2
    nele = noA + noB
    occ = all_dets(sorb, noA, noB)
    h1, h2 = synth_integrals(sorb)
    onv = m1.tensor_to_onv(T(occ), sorb)
    comb, hm = m1.get_comb_hij_fused(onv, h1, h2, sorb, nele, noA, noB)
    comb32, hm32 = m1.get_comb_hij_fused(onv, h1.float(), h2.float(), sorb, nele, noA, noB)
    comb_u, pm = m1.get_comb_tensor(onv, sorb, nele, noA, noB, True)
    h3d = m1.get_hij_torch(onv, comb, h1, h2, sorb, nele)
    h2d = m1.get_hij_torch(onv, onv, h1, h2, sorb, nele)
    h2d32 = m1.get_hij_torch(onv, onv, h1.float(), h2.float(), sorb, nele)
    assert torch.equal(comb, comb32) and torch.equal(comb, comb_u) and torch.equal(h3d, hm)
    pm1 = m1.onv_to_tensor(onv, sorb)
    torch.set_default_dtype(torch.float32)
    pm1_f32 = m1.onv_to_tensor(onv, sorb)
    torch.set_default_dtype(torch.float64)
    np.savez_compressed(f"{HERE}/c1_sorb8_all36.npz", sorb=sorb, noA=noA, noB=noB, occ=occ, onv=onv.numpy(),
                        h1e=h1.numpy(), h2e=h2.numpy(), comb=comb.numpy(), hmat=hm.numpy(), hmat_f32=hm32.numpy(),
                        comb_pm1=pm.numpy(), hij2d=h2d.numpy(), hij2d_f32=h2d32.numpy(), pm1=pm1.numpy(),
                        pm1_f32=pm1_f32.numpy())

    # ---- 2. asymmetric occupations (slot interleave + the `idx % noAA` quirk) -------------------
    out = {}
    for (sorb, noA, noB) in [(12, 3, 2), (12, 2, 4), (16, 5, 3), (10, 1, 1), (10, 4, 4), (4, 1, 1), (6, 2, 1)]:
        nele = noA + noB
        occ = rand_occ(4, sorb, noA, noB, seed=100 + sorb + noA)
        h1, h2 = synth_integrals(sorb)
        onv = m1.tensor_to_onv(T(occ), sorb)
        comb, hm = m1.get_comb_hij_fused(onv, h1, h2, sorb, nele, noA, noB)
        _, hm32 = m1.get_comb_hij_fused(onv, h1.float(), h2.float(), sorb, nele, noA, noB)
        key = f"s{sorb}_a{noA}_b{noB}"
        out[key + "_onv"] = onv.numpy(); out[key + "_comb"] = comb.numpy()
        out[key + "_hmat"] = hm.numpy(); out[key + "_hmat_f32"] = hm32.numpy()
    np.savez_compressed(f"{HERE}/asym_small.npz", **out)

    # ---- 3. word-boundary cases, bra_len 1/2/3: sampled ranks ------------------------------------
    out = {}
    for (sorb, noA, noB) in [(64, 6, 5), (66, 4, 3), (126, 3, 3), (128, 4, 4), (130, 3, 4), (184, 3, 3), (192, 4, 3),
                             (120, 6, 6)]:
        L = (sorb - 1) // 64 + 1
        m = mods[L][0]
        nele = noA + noB
        occ = rand_occ(3, sorb, noA, noB, seed=200 + sorb)
        # make one walker touch the top orbitals and word boundaries
        occ[0] = 0; occ[0, [sorb - 2 * (i + 1) for i in range(noA)]] = 1; occ[0, [sorb - 2 * i - 1 for i in range(noB)]] = 1
        h1, h2 = synth_integrals(sorb)
        onv = m.tensor_to_onv(T(occ), sorb)
        comb, hm = m.get_comb_hij_fused(onv, h1, h2, sorb, nele, noA, noB)
        _, hm32 = m.get_comb_hij_fused(onv, h1.float(), h2.float(), sorb, nele, noA, noB)
        ncomb = comb.shape[1]
        k = sorb // 2
        nvA, nvB = k - noA, k - noB
        d = np.cumsum([noA * nvA, noB * nvB, noA * (noA - 1) // 2 * (nvA * (nvA - 1) // 2),
                       noB * (noB - 1) // 2 * (nvB * (nvB - 1) // 2)])
        edges = sorted({0, 1, ncomb - 1} | {int(x) + 1 for x in d} | {int(x) for x in d})
        g = np.random.default_rng(300 + sorb)
        ranks = np.unique(np.concatenate([np.array(edges), g.integers(0, ncomb, 256)]))
        ranks = ranks[ranks < ncomb]
        key = f"s{sorb}_a{noA}_b{noB}"
        out[key + "_onv"] = onv.numpy(); out[key + "_ranks"] = ranks
        out[key + "_comb"] = comb.numpy()[:, ranks]; out[key + "_hmat"] = hm.numpy()[:, ranks]
        out[key + "_hmat_f32"] = hm32.numpy()[:, ranks]
        out[key + "_rowsum"] = hm.numpy().sum(1); out[key + "_rowabs"] = np.abs(hm.numpy()).sum(1)
        out[key + "_comb_sha"] = np.array(sha(comb.numpy())); out[key + "_hmat_sha"] = np.array(sha(hm.numpy()))
        out[key + "_pm1"] = m.onv_to_tensor(onv, sorb).numpy()
        # generic pair path: 2-D matrix among a subset of kets of walker 0 (degree 0/1/2/>2 all occur)
        sub = comb[0, torch.from_numpy(ranks[:48])].contiguous()
        out[key + "_hij2d"] = m.get_hij_torch(sub, sub, h1, h2, sorb, nele).numpy()
    np.savez_compressed(f"{HERE}/word_boundary.npz", **out)

    # ---- 4. the shipped Fe2S2 problem (example/Fe2S2/fe2s2-OO.pth: sorb 40, 15a15b) -------------
    e = load_fe2s2()
    sorb, nele, noA, noB = int(e["sorb"]), int(e["nele"]), int(e["noa"]), int(e["nob"])
    h1, h2 = e["h1e"].double().contiguous(), e["h2e"].double().contiguous()
    ci = e["ci_space"].contiguous()
    comb, hm = m1.get_comb_hij_fused(ci[:64].contiguous(), h1, h2, sorb, nele, noA, noB)
    _, hm32 = m1.get_comb_hij_fused(ci[:64].contiguous(), h1.float(), h2.float(), sorb, nele, noA, noB)
    np.savez_compressed(f"{HERE}/fe2s2_inputs.npz", sorb=sorb, nele=nele, noA=noA, noB=noB, h1e=h1.numpy(),
                        h2e=h2.numpy(), ci_space=ci.numpy(), ecore=float(e["ecore"]), e_ref=float(e["e_lst"][0]))
    np.savez_compressed(f"{HERE}/fe2s2_hmat.npz", hmat8=hm[:8].numpy(), hmat8_f32=hm32[:8].numpy(),
                        comb8_sha=np.array(sha(comb[:8].numpy())),
                        rowsum=hm.numpy().sum(1), rowabs=np.abs(hm.numpy()).sum(1),
                        comb_sha=np.array([sha(comb[i].numpy()) for i in range(64)]),
                        hmat_sha=np.array([sha(hm[i].numpy()) for i in range(64)]))

    # ---- 5. wavefunction_lut (libs/C_extension.pyi:305-355 + random multi-word) -----------------
    out = {}
    for (sorb_l, noA_l, noB_l) in [(40, 15, 15), (100, 5, 6), (184, 3, 3)]:
        L = (sorb_l - 1) // 64 + 1
        m = mods[L][0]
        occ = np.unique(rand_occ(400, sorb_l, noA_l, noB_l, seed=500 + sorb_l), axis=0)
        keys = m.tensor_to_onv(T(occ), sorb_l)
        w = keys.numpy().view(np.uint64)
        order = np.lexsort(tuple(w[:, k] for k in range(w.shape[1])))
        keys = keys[torch.from_numpy(order)].contiguous()
        q = torch.cat([keys[::3], m.tensor_to_onv(T(rand_occ(100, sorb_l, noA_l, noB_l, seed=900 + sorb_l)), sorb_l)]).contiguous()
        idx, mask = m.wavefunction_lut(keys, q, sorb_l)
        key = f"s{sorb_l}"
        out[key + "_keys"] = keys.numpy(); out[key + "_query"] = q.numpy()
        out[key + "_idx"] = idx.numpy(); out[key + "_mask"] = mask.numpy()
    np.savez_compressed(f"{HERE}/wavefunction_lut.npz", **out)

    # ---- 6. end-to-end local energy through the reference's Python (vmc/energy/eloc.py) ---------
    stub = os.path.join(args.scratch, "pyharness")
    os.makedirs(os.path.join(stub, "libs"), exist_ok=True)
    open(os.path.join(stub, "libs", "__init__.py"), "w").close()
    shutil.copy(mods[1][1], os.path.join(stub, "libs", os.path.basename(mods[1][1])))
    with open(os.path.join(stub, "loguru.py"), "w") as f:  # inert: logging only
        f.write("class _L:\n    def __getattr__(self, k):\n        return lambda *a, **kw: None\nlogger = _L()\n")
    with open(os.path.join(stub, "jaxtyping.py"), "w") as f:  # inert: type annotations only
        f.write("class _M(type):\n    def __getitem__(c, k):\n        return c\n"
                "class _B(metaclass=_M):\n    pass\n"
                "Float = Int = UInt8 = Bool = Complex = Shaped = Num = Integer = Real = Inexact = Array = _B\n"
                "def __getattr__(name):\n    return _B\n")
    sys.path.insert(0, REF)
    sys.path.insert(0, stub)
    from functools import partial

    from utils.public_function import WavefunctionLUT, ansatz_batch
    from utils.stats.mc_stats import operator_statistics
    from vmc.ansatz.rbm.rbm import RBMWavefunction
    from vmc.energy.eloc import local_energy

    x = ci[:32].contiguous()
    rbm = RBMWavefunction(sorb, alpha=2, iscale=0.001, rbm_type="real")
    g = torch.Generator().manual_seed(7)
    W = 0.01 * (torch.rand(2 * sorb, sorb, generator=g, dtype=torch.float64) - 0.5)
    hb = 0.01 * (torch.rand(2 * sorb, generator=g, dtype=torch.float64) - 0.5)
    vb = 0.1 * (torch.rand(sorb, generator=g, dtype=torch.float64) - 0.5)
    rbm.init(hb, W, vb)

    def _ab(x, func):
        return ansatz_batch(func, x, 100000, sorb, torch.device("cpu"), torch.double)

    eloc_s, _, psi_s, _ = local_energy(x, h1, h2, rbm, _ab, sorb, nele, noA, noB, dtype=torch.double, use_unique=True)
    eloc_r, _, psi_r, _ = local_energy(x, h1, h2, rbm, _ab, sorb, nele, noA, noB, dtype=torch.double, use_unique=True,
                                       reduce_psi=True, eps=1e-2, eps_sample=0)
    # sample-space: LUT over the first 4096 determinants of ci_space with the RBM's amplitudes
    keys = ci[:4096].contiguous()
    with torch.no_grad():
        wf = _ab(keys, rbm)
    lut = WavefunctionLUT(keys, wf, sorb, device="cpu")
    eloc_ss, _, psi_ss, _ = local_energy(x, h1, h2, rbm, _ab, sorb, nele, noA, noB, dtype=torch.double, WF_LUT=lut,
                                         use_sample_space=True, index=(0, 32))
    # complex LUT values (BDG-RNN style amplitudes): same keys, synthetic complex psi
    wfc = torch.complex(wf, 0.3 * wf.flip(0))
    lutc = WavefunctionLUT(keys, wfc, sorb, device="cpu")
    eloc_ssc, _, psi_ssc, _ = local_energy(x, h1, h2, rbm, _ab, sorb, nele, noA, noB, dtype=torch.complex128,
                                           WF_LUT=lutc, use_sample_space=True, index=(0, 32))
    # LUT-assisted REDUCE (Func: lookup hit/miss split + unique)
    lut_small = WavefunctionLUT(ci[:512].contiguous(), wf[:512], sorb, device="cpu")
    eloc_rl, _, _, _ = local_energy(x, h1, h2, rbm, _ab, sorb, nele, noA, noB, dtype=torch.double, use_unique=True,
                                    WF_LUT=lut_small, reduce_psi=True, eps=1e-2, eps_sample=0)
    prob = torch.rand(32, generator=g, dtype=torch.float64); prob = prob / prob.sum()
    st = operator_statistics(eloc_s.detach(), prob, 1000, "E")
    np.savez_compressed(f"{HERE}/eloc_e2e_fe2s2.npz", x=x.numpy(), W=W.numpy(), hb=hb.numpy(), vb=vb.numpy(),
                        psi_lut_keys=keys.numpy(), psi_lut=wf.numpy(), psi_lut_c=wfc.numpy(),
                        eloc_simple=eloc_s.numpy(), psi_simple=psi_s.detach().numpy(),
                        eloc_reduce=eloc_r.numpy(), psi_reduce=psi_r.detach().numpy(),
                        eloc_reduce_lut=eloc_rl.numpy(),
                        eloc_sample_space=eloc_ss.numpy(), psi_sample_space=psi_ss.numpy(),
                        eloc_sample_space_c=eloc_ssc.numpy(), psi_sample_space_c=psi_ssc.numpy(),
                        prob=prob.numpy(), stat_counts=1000, stat_mean=st["mean"].numpy(), stat_var=st["var"].numpy(),
                        stat_sd=st["sd"].numpy(), stat_se=st["se"].numpy())
    print("stats object:", st)
    tot = sum(os.path.getsize(p) for p in glob.glob(f"{HERE}/*.npz"))
    print(f"golden written: {tot/2**20:.2f} MiB")


if __name__ == "__main__":
    main()

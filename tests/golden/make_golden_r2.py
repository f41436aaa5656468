#!/usr/bin/env python3
"""Second batch of golden vectors, captured from the REFERENCE's own Python running on its own CPU extension
(development container only; see make_golden.py for how the reference is built and imported).

What is captured (all on the shipped Fe2S2 problem, ci_space[:32], fixed-weight amplitudes):
  eloc_flip_multipsi_fe2s2.npz  vmc/energy/flip.py:66-418 (_simple_flip / _reduce_psi_flip / _only_sample_space_flip) and
                                the use_multi_psi branches of vmc/energy/eloc.py:134-401
  eloc_complex_module.npz       SIMPLE / REDUCE (+ LUT, + spin-flip) with a complex128 module (pynqs_amd.rbm.ComplexRBM as the input amplitude)
  grad_fe2s2.npz                vmc/grad/energy_grad.py:118-184 (`grad`) under DistributedDataParallel, world size 1 and 2 (gloo)
  gfmc_fe2s2.npz                gfmc/walker.py:167-235 (_calculate_green_kernel), :260-279 (sample_update), :340-408 (branching, ws 1 and 2)
  eloc_rbm_flavours.npz         SIMPLE / REDUCE with RBMWavefunction(rbm_type = "tanh" / "pRBM" / "cos") (vmc/ansatz/rbm/rbm.py:199-211)
  eloc_spin_raising_fe2s2.npz   use_spin_raising (<S-S+> with the integrals of utils/pyscf_helper/operator.py:93-137): SIMPLE / REDUCE / SAMPLE_SPACE
                                local_energy and total_energy's REDUCE + sample-space form (etot.py:93-142)
  ansatz_helpers.npz            permute_sgn / constrain_make_charts of the compiled reference extension
  sampler_merge.npz             merge_rank_sample (cpp_src/tensor/cpu_tensor.cpp:537-556) and Sampler.gather_scatter_sample
                                (vmc/sample.py:627-772) run by two gloo ranks, both `use_same_tree` settings
Only DATA is written: inputs and the reference's outputs.

usage: python tests/golden/make_golden_r2.py [--scratch /tmp/refbuild]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402

REF = MG.REF


def harness(scratch: str):
    """sys.path for the reference's Python: [stub dir with libs/C_extension.so + inert loguru/jaxtyping, reference]."""
    import shutil

    mod, so = MG.build_ref(scratch, 1)
    stub = os.path.join(scratch, "pyharness")
    os.makedirs(os.path.join(stub, "libs"), exist_ok=True)
    open(os.path.join(stub, "libs", "__init__.py"), "w").close()
    dst = os.path.join(stub, "libs", os.path.basename(so))
    if not os.path.exists(dst):
        shutil.copy(so, dst)
    if not os.path.exists(os.path.join(stub, "loguru.py")):
        with open(os.path.join(stub, "loguru.py"), "w") as f:
            f.write("class _L:\n    def __getattr__(self, k):\n        return lambda *a, **kw: None\nlogger = _L()\n")
        with open(os.path.join(stub, "jaxtyping.py"), "w") as f:
            f.write("class _M(type):\n    def __getitem__(c, k):\n        return c\n"
                    "class _B(metaclass=_M):\n    pass\n"
                    "Float = Int = UInt8 = Bool = Complex = Shaped = Num = Integer = Real = Inexact = Array = _B\n"
                    "def __getattr__(name):\n    return _B\n")
    for p in (REF, stub):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REF)
    sys.path.insert(0, stub)
    return mod


def load_inputs():
    f = np.load(f"{HERE}/fe2s2_inputs.npz")
    e = np.load(f"{HERE}/eloc_e2e_fe2s2.npz")
    T = torch.from_numpy
    return dict(sorb=int(f["sorb"]), nele=int(f["nele"]), noA=int(f["noA"]), noB=int(f["noB"]), h1e=T(f["h1e"]), h2e=T(f["h2e"]),
                ci=T(f["ci_space"]), x=T(e["x"]), W=T(e["W"]), hb=T(e["hb"]), vb=T(e["vb"]))


def second_rbm_params(sorb: int):
    """Weights of the `extra` factor f(x) of the multi-psi forms (real RBM, alpha = 1) and of the complex128 module."""
    g = torch.Generator().manual_seed(11)
    W2 = 0.02 * (torch.rand(sorb, sorb, generator=g, dtype=torch.float64) - 0.5)
    hb2 = 0.02 * (torch.rand(sorb, generator=g, dtype=torch.float64) - 0.5)
    vb2 = 0.05 * (torch.rand(sorb, generator=g, dtype=torch.float64) - 0.5)
    g = torch.Generator().manual_seed(13)
    Wc = 0.02 * (torch.rand(sorb, sorb, 2, generator=g, dtype=torch.float64) - 0.5)
    hbc = 0.02 * (torch.rand(sorb, 2, generator=g, dtype=torch.float64) - 0.5)
    vbc = 0.05 * (torch.rand(sorb, 2, generator=g, dtype=torch.float64) - 0.5)
    return (W2, hb2, vb2), (Wc, hbc, vbc)


def complex_module(Wc, hbc, vbc):
    """The complex128 amplitude (an INPUT of the fixtures, defined in this repository: pynqs_amd/rbm.py ComplexRBM; the
    reference's rbm_type="complex" raises in psi(), rbm.py:198-205).  Loaded by path so that the package (and its native
    library) is not imported into the reference's process."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_amp_rbm", os.path.join(os.path.dirname(os.path.dirname(HERE)), "pynqs_amd", "rbm.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.ComplexRBM(Wc, hbc, vbc)


class Holder:
    """What the reference's multi-psi code dereferences: ansatz.module.sample / ansatz.module.extra (a DDP-wrapped model)."""

    def __init__(self, sample, extra):
        self.module = types.SimpleNamespace(sample=sample, extra=extra)

    def __call__(self, x):
        return self.module.sample(x) * self.module.extra(x)


# --------------------------------------------------------------------------------------------------------------
def section_eloc(I, out_dir):
    from utils.public_function import SpinProjection, WavefunctionLUT, ansatz_batch, spin_flip_onv
    from vmc.ansatz.rbm.rbm import RBMWavefunction
    from vmc.energy.eloc import local_energy

    sorb, nele, noA, noB = I["sorb"], I["nele"], I["noA"], I["noB"]
    h1, h2, x, ci = I["h1e"], I["h2e"], I["x"], I["ci"]
    (W2, hb2, vb2), (Wc, hbc, vbc) = second_rbm_params(sorb)
    rbm = RBMWavefunction(sorb, alpha=2, rbm_type="real"); rbm.init(I["hb"].clone(), I["W"].clone(), I["vb"].clone())
    extra = RBMWavefunction(sorb, alpha=1, rbm_type="real"); extra.init(hb2.clone(), W2.clone(), vb2.clone())
    crbm = complex_module(Wc, hbc, vbc)
    multi = Holder(rbm, extra)
    cmulti = Holder(rbm, crbm)
    SpinProjection.init(nele, 0)
    eta = SpinProjection.eta
    extra_norm = torch.tensor(1.3, dtype=torch.float64)
    # the extra factor f is an RBM amplitude of size ~2^40: normalise |f|^2 to O(1) so that the 1e-8 Ha bound means something
    extra_norm_m = torch.tensor(1.3 * 2.0**40, dtype=torch.float64)

    def ab(dt):
        return lambda x, func: ansatz_batch(func, x, 100000, sorb, torch.device("cpu"), dt)

    def le(ansatz, dt=torch.double, **kw):
        e, s, p, _ = local_energy(x, h1, h2, ansatz, ab(dt), sorb, nele, noA, noB, dtype=dt, **kw)
        return e.detach().numpy(), p.detach().numpy()

    # look-up tables: keys closed under the alpha<->beta swap so that the projected sample-space form finds partners
    base = ci[:2048].contiguous()
    keys = torch.unique(torch.cat([base, spin_flip_onv(base, sorb)]), dim=0)
    with torch.no_grad():
        wf = ab(torch.double)(keys, rbm)
    lut = WavefunctionLUT(keys, wf, sorb, device="cpu")
    wfc = torch.complex(wf, 0.3 * wf.flip(0))
    lutc = WavefunctionLUT(keys, wfc, sorb, device="cpu")
    lut_small = WavefunctionLUT(keys[:700].contiguous(), wf[:700], sorb, device="cpu")

    out = dict(x=x.numpy(), eta=eta, extra_norm=float(extra_norm), extra_norm_multi=float(extra_norm_m), W2=W2.numpy(), hb2=hb2.numpy(), vb2=vb2.numpy(),
               Wc=Wc.numpy(), hbc=hbc.numpy(), vbc=vbc.numpy(), lut_keys=keys.numpy(), lut_wf=wf.numpy(), lut_wfc=wfc.numpy(),
               n_lut_small=700)

    def put(name, ep):
        out["eloc_" + name], out["psi_" + name] = ep

    # ---- spin-flip projected forms (flip.py) ----------------------------------------------------------------
    put("simple_flip", le(rbm, use_spin_flip=True, extra_norm=extra_norm))
    put("reduce_flip", le(rbm, use_spin_flip=True, extra_norm=extra_norm, reduce_psi=True, eps=1e-2, eps_sample=0))
    put("reduce_flip_lut", le(rbm, use_spin_flip=True, extra_norm=extra_norm, reduce_psi=True, eps=1e-2, eps_sample=0, WF_LUT=lut_small))
    put("ss_flip", le(rbm, use_spin_flip=True, extra_norm=extra_norm, use_sample_space=True, WF_LUT=lut, index=(0, 32)))
    put("ss_flip_c", le(rbm, torch.complex128, use_spin_flip=True, extra_norm=extra_norm, use_sample_space=True, WF_LUT=lutc, index=(0, 32)))
    # ---- multi-psi forms without projection (eloc.py) -------------------------------------------------------
    put("simple_multi", le(multi, use_multi_psi=True, extra_norm=extra_norm_m))
    put("reduce_multi", le(multi, use_multi_psi=True, extra_norm=extra_norm_m, reduce_psi=True, eps=1e-2, eps_sample=0))
    put("ss_multi", le(multi, use_multi_psi=True, extra_norm=extra_norm_m, use_sample_space=True, WF_LUT=lut, index=(0, 32)))
    put("simple_multi_c", le(cmulti, torch.complex128, use_multi_psi=True, extra_norm=extra_norm_m))
    # ---- multi-psi + projection (flip.py) ---------------------------------------------------------------------
    put("simple_flip_multi", le(multi, use_spin_flip=True, use_multi_psi=True, extra_norm=extra_norm_m))
    put("reduce_flip_multi", le(multi, use_spin_flip=True, use_multi_psi=True, extra_norm=extra_norm_m, reduce_psi=True, eps=1e-2, eps_sample=0))
    put("ss_flip_multi", le(multi, use_spin_flip=True, use_multi_psi=True, extra_norm=extra_norm_m, use_sample_space=True, WF_LUT=lut, index=(0, 32)))
    put("ss_flip_multi_c", le(cmulti, torch.complex128, use_spin_flip=True, use_multi_psi=True, extra_norm=extra_norm_m, use_sample_space=True,
                              WF_LUT=lutc, index=(0, 32)))
    np.savez_compressed(f"{out_dir}/eloc_flip_multipsi_fe2s2.npz", **out)

    # ---- complex128 module on the generic SIMPLE / REDUCE path (C4's dtype) -----------------------------------
    c = dict(x=x.numpy(), Wc=Wc.numpy(), hbc=hbc.numpy(), vbc=vbc.numpy())
    c["eloc_simple"], c["psi_simple"] = le(crbm, torch.complex128)
    c["eloc_reduce"], c["psi_reduce"] = le(crbm, torch.complex128, reduce_psi=True, eps=1e-2, eps_sample=0)
    with torch.no_grad():
        wfk = ab(torch.complex128)(keys[:700].contiguous(), crbm)
    lutk = WavefunctionLUT(keys[:700].contiguous(), wfk, sorb, device="cpu")
    c["lut_keys"], c["lut_wf"] = keys[:700].numpy(), wfk.numpy()
    c["eloc_reduce_lut"], _ = le(crbm, torch.complex128, reduce_psi=True, eps=1e-2, eps_sample=0, WF_LUT=lutk)
    c["eloc_simple_flip"], c["psi_simple_flip"] = le(crbm, torch.complex128, use_spin_flip=True, extra_norm=extra_norm)
    c["eta"], c["extra_norm"] = eta, float(extra_norm)
    np.savez_compressed(f"{out_dir}/eloc_complex_module.npz", **c)
    print("eloc sections done:", {k: v.shape for k, v in out.items() if k.startswith("eloc_")})


# --------------------------------------------------------------------------------------------------------------
def section_rbm_flavours(I, out_dir):
    """SIMPLE and REDUCE local energies with the reference's other real-parameter RBM amplitudes (rbm.py:199-211: "tanh", "pRBM",
    "cos"), same weights as eloc_e2e_fe2s2.npz."""
    from utils.public_function import ansatz_batch
    from vmc.ansatz.rbm.rbm import RBMWavefunction
    from vmc.energy.eloc import local_energy

    sorb, nele, noA, noB = I["sorb"], I["nele"], I["noA"], I["noB"]
    out = dict(x=I["x"].numpy())
    for kind in ("tanh", "pRBM", "cos"):
        dt = torch.complex128 if kind == "pRBM" else torch.double
        m = RBMWavefunction(sorb, alpha=2, rbm_type=kind)
        m.init(I["hb"].clone(), I["W"].clone(), I["vb"].clone())
        ab = lambda x, func: ansatz_batch(func, x, 100000, sorb, torch.device("cpu"), dt)  # noqa: E731
        for tag, kw in (("simple", {}), ("reduce", dict(reduce_psi=True, eps=1e-2, eps_sample=0))):
            e, _, p, _ = local_energy(I["x"], I["h1e"], I["h2e"], m, ab, sorb, nele, noA, noB, dtype=dt, **kw)
            out[f"eloc_{tag}_{kind}"], out[f"psi_{tag}_{kind}"] = e.detach().numpy(), p.detach().numpy()
    np.savez_compressed(f"{out_dir}/eloc_rbm_flavours.npz", **out)
    print("rbm flavours:", {k: (v.dtype, float(np.abs(v).max())) for k, v in out.items() if k.startswith("eloc_")})


# --------------------------------------------------------------------------------------------------------------
def section_spin_raising(I, out_dir):
    """<S-S+> next to the energy (use_spin_raising, eloc.py:173-188,250-310,377-400; etot.py:93-142): the S-S+ integrals of
    utils/pyscf_helper/operator.py:93-137 (spin_raising) are part of the fixture."""
    from utils.public_function import WavefunctionLUT, ansatz_batch
    from utils.pyscf_helper.operator import spin_raising
    from vmc.ansatz.rbm.rbm import RBMWavefunction
    from vmc.energy.eloc import local_energy
    from vmc.energy.etot import total_energy

    sorb, nele, noA, noB = I["sorb"], I["nele"], I["noA"], I["noB"]
    h1, h2, x, ci = I["h1e"], I["h2e"], I["x"], I["ci"]
    h1s, h2s = spin_raising(sorb, c1=1.0)
    rbm = RBMWavefunction(sorb, alpha=2, rbm_type="real"); rbm.init(I["hb"].clone(), I["W"].clone(), I["vb"].clone())
    ab = lambda x_, func: ansatz_batch(func, x_, 100000, sorb, torch.device("cpu"), torch.double)  # noqa: E731
    keys = ci[:2048].contiguous()
    with torch.no_grad():
        wf = ab(keys, rbm)
    lut = WavefunctionLUT(keys, wf, sorb, device="cpu")
    out = dict(x=x.numpy(), h1e_spin=h1s.numpy(), h2e_spin=h2s.numpy(), lut_keys=keys.numpy(), lut_wf=wf.numpy())
    for tag, kw in (("simple", {}), ("reduce", dict(reduce_psi=True, eps=1e-2, eps_sample=0)),
                    ("ss", dict(use_sample_space=True, WF_LUT=lut, index=(0, 32)))):
        e, sl, p, _ = local_energy(x, h1, h2, rbm, ab, sorb, nele, noA, noB, use_spin_raising=True, h1e_spin=h1s, h2e_spin=h2s, **kw)
        out[f"eloc_{tag}"], out[f"sloc_{tag}"], out[f"psi_{tag}"] = e.detach().numpy(), sl.detach().numpy(), p.detach().numpy()
    # the production form: REDUCE for the energy, <S-S+> recomputed in the sample space (etot.py:119-142)
    e, sl, _ = total_energy(x, 16, 100000, h1, h2, rbm, sorb, nele, noA, noB, WF_LUT=lut, use_spin_raising=True, h1e_spin=h1s, h2e_spin=h2s,
                            reduce_psi=True, eps=1e-2, eps_sample=0)
    out["eloc_etot_reduce"], out["sloc_etot_reduce"] = e.detach().numpy(), sl.detach().numpy()
    np.savez_compressed(f"{out_dir}/eloc_spin_raising_fe2s2.npz", **out)
    print("spin raising:", {k: float(np.abs(v).max()) for k, v in out.items() if k.startswith("sloc_")})


# --------------------------------------------------------------------------------------------------------------
def _grad_case(I, kind, rank, ws, AD_MAX_DIM, use_pow):
    """Runs the reference's grad() on this rank's shard of 32 walkers; returns the parameter gradients after the DDP
    reduction and this rank's loss."""
    from torch.nn.parallel import DistributedDataParallel as DDP
    from libs.C_extension import onv_to_tensor
    from vmc.ansatz.rbm.rbm import RBMWavefunction
    from vmc.grad.energy_grad import grad

    sorb = I["sorb"]
    (W2, hb2, vb2), (Wc, hbc, vbc) = second_rbm_params(sorb)
    if kind == "real":
        m = RBMWavefunction(sorb, alpha=2, rbm_type="real"); m.init(I["hb"].clone(), I["W"].clone(), I["vb"].clone())
        dt = torch.double
    else:
        m = complex_module(Wc, hbc, vbc)
        dt = torch.complex128
    nqs = DDP(m)
    g = torch.Generator().manual_seed(21)
    n = 32
    prob = torch.rand(n, generator=g, dtype=torch.float64); prob = prob / prob.sum()
    er = torch.rand(n, generator=g, dtype=torch.float64) - 108.0
    ei = 0.1 * (torch.rand(n, generator=g, dtype=torch.float64) - 0.5)
    eloc = er if kind == "real" else torch.complex(er, ei)
    powr = 0.5 + torch.rand(n, generator=g, dtype=torch.float64)
    e_total = (eloc * prob).sum()
    k, res = divmod(n, ws)
    b = rank * k + min(rank, res)
    e = b + k + (1 if rank < res else 0)
    states = onv_to_tensor(I["x"][b:e].contiguous(), sorb)
    # the sampler hands every rank prob * world_size (vmc/sample.py:772)
    grad(nqs, states, prob[b:e] * ws, eloc[b:e], e_total, powr[b:e] if use_pow else 1.0, dt, AD_MAX_DIM)
    gr = {name: p.grad.detach().clone().numpy() for name, p in m.named_parameters()}
    return dict(prob=prob.numpy(), eloc=eloc.numpy(), pow=powr.numpy(), e_total=np.asarray(e_total.numpy()), grads=gr)


def _dist_worker(rank, ws, port, scratch, q):
    import torch.distributed as dist

    torch.set_default_dtype(torch.float64)
    torch.set_num_threads(2)
    harness(scratch)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    I = load_inputs()
    res = {}
    for kind in ("real", "complex"):
        for amd, use_pow in ((-1, False), (5, True)):
            r = _grad_case(I, kind, rank, ws, amd, use_pow)
            res[f"grad_{kind}_amd{amd}_pow{int(use_pow)}"] = r
    # ---- GFMC branching (walker.py:340-408) ---------------------------------------------------------------
    from gfmc.walker import GFMC

    sorb = I["sorb"]
    n = 24  # equal shards: the reference's slot offset (walker.py:369) is only right for equal shard sizes
    k = n // ws
    xs = I["ci"][100:100 + n].contiguous()
    g = torch.Generator().manual_seed(31)
    w = torch.rand(n, generator=g, dtype=torch.float64) + 0.05
    w[3] = 4.0; w[17] = 0.0
    me = types.SimpleNamespace(device=torch.device("cpu"), world_size=ws, rank=rank)
    torch.manual_seed(1000 + rank)
    xi = torch.rand(k)  # the draw branching() is about to make (same generator state)
    torch.manual_seed(1000 + rank)
    xb = GFMC.branching(me, xs[rank * k:(rank + 1) * k].contiguous(), w[rank * k:(rank + 1) * k].contiguous())
    res["branch"] = dict(x=xs.numpy(), w=w.numpy(), xi=xi.numpy(), out=xb.numpy())
    # ---- Sampler.gather_scatter_sample (vmc/sample.py:627-772) -------------------------------------------------
    from libs.C_extension import onv_to_tensor
    from vmc.sample import Sampler

    ci = I["ci"]
    for same_tree in (True, False):
        # every rank's own distinct samples; with use_same_tree=False the ranks overlap
        if same_tree:
            lo = [0, 37, 37 + 41, 37 + 41 + 29]
            mine = ci[200 + lo[rank]:200 + lo[rank + 1]]
        else:
            gg = torch.Generator().manual_seed(41 + rank)
            mine = ci[300:360][torch.randperm(60, generator=gg)[:35 + 6 * rank]]
        mine = mine.contiguous()
        occ = ((onv_to_tensor(mine, sorb) + 1) / 2).to(torch.uint8)  # 0/1 occupations as the sampler produces them
        gg = torch.Generator().manual_seed(51 + rank)
        counts = torch.randint(1, 50, (mine.size(0),), generator=gg)
        # psi must be a function of the determinant (duplicates across ranks carry the same value)
        wf = torch.complex(mine.double().sum(1) / 100.0, mine[:, 0].double() / 50.0)
        me = types.SimpleNamespace(sorb=sorb, device=torch.device("cpu"), world_size=ws, rank=rank, use_LUT=True,
                                   use_same_tree=same_tree, dtype=torch.complex128, all_sample_counts=None)
        ur, _, pr, lutr = Sampler.gather_scatter_sample(me, occ, counts, wf)
        res[f"gs_tree{int(same_tree)}"] = dict(occ=occ.numpy(), counts=counts.numpy(), wf=wf.numpy(), unique_rank=ur.numpy(),
                                                prob_rank=pr.numpy(), lut_keys=lutr.bra_key.numpy(), lut_wf=lutr.wf_value.numpy(),
                                                all_counts=(me.all_sample_counts.numpy() if me.all_sample_counts is not None else np.zeros(0)))
    # ---- statistics (utils/stats) at this world size -------------------------------------------------------------
    from utils.stats.mc_stats import operator_statistics

    g = torch.Generator().manual_seed(61)
    n = 33
    prob = torch.rand(n, generator=g, dtype=torch.float64); prob = prob / prob.sum()
    el = torch.complex(torch.rand(n, generator=g, dtype=torch.float64) - 108.0, 0.1 * torch.rand(n, generator=g, dtype=torch.float64))
    kk, rr = divmod(n, ws)
    b = rank * kk + min(rank, rr); e = b + kk + (1 if rank < rr else 0)
    st = operator_statistics(el[b:e], prob[b:e] * ws, 4000, "E")
    res["stats"] = dict(prob=prob.numpy(), eloc=el.numpy(), mean=np.asarray(st["mean"]), var=np.asarray(st["var"]), sd=np.asarray(st["sd"]),
                        se=np.asarray(st["se"]))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def section_dist(scratch, out_dir):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    allres = {}
    for ws, port in ((1, 29611), (2, 29612)):
        q = ctx.Queue()
        ps = [ctx.Process(target=_dist_worker, args=(r, ws, port, scratch, q)) for r in range(ws)]
        [p.start() for p in ps]
        got = dict(q.get(timeout=600) for _ in range(ws))
        [p.join() for p in ps]
        assert all(p.exitcode == 0 for p in ps)
        allres[ws] = got
    # ---- grad -----------------------------------------------------------------------------------------------
    out = {}
    for key in allres[1][0]:
        if not key.startswith("grad_"):
            continue
        r1 = allres[1][0][key]
        for f in ("prob", "eloc", "pow", "e_total"):
            out[f"{key}_{f}"] = r1[f]
        for name, gv in r1["grads"].items():
            out[f"{key}_ws1_{name}"] = gv
        # after DDP's all-reduce (mean over ranks) both ranks hold the same gradient
        for name, gv in allres[2][0][key]["grads"].items():
            assert np.array_equal(gv, allres[2][1][key]["grads"][name])
            out[f"{key}_ws2_{name}"] = gv
    np.savez_compressed(f"{out_dir}/grad_fe2s2.npz", **out)
    # ---- branching ------------------------------------------------------------------------------------------
    b = dict(x=allres[1][0]["branch"]["x"], w=allres[1][0]["branch"]["w"])
    b["ws1_xi"], b["ws1_out"] = allres[1][0]["branch"]["xi"], allres[1][0]["branch"]["out"]
    for r in (0, 1):
        b[f"ws2_xi_r{r}"], b[f"ws2_out_r{r}"] = allres[2][r]["branch"]["xi"], allres[2][r]["branch"]["out"]
    # ---- sampler merge --------------------------------------------------------------------------------------
    s = {}
    for t in (0, 1):
        for r in (0, 1):
            for f, v in allres[2][r][f"gs_tree{t}"].items():
                s[f"tree{t}_r{r}_{f}"] = v
    # ---- stats ---------------------------------------------------------------------------------------------------
    st = {}
    for ws in (1, 2):
        for f, v in allres[ws][0]["stats"].items():
            st[f"ws{ws}_{f}"] = v
    return b, s, st


def section_gfmc(I, out_dir, branch):
    from gfmc.walker import GFMC
    from utils.public_function import WavefunctionLUT, ansatz_batch
    from vmc.ansatz.rbm.rbm import RBMWavefunction

    sorb, nele, noA, noB = I["sorb"], I["nele"], I["noA"], I["noB"]
    rbm = RBMWavefunction(sorb, alpha=2, rbm_type="real"); rbm.init(I["hb"].clone(), I["W"].clone(), I["vb"].clone())
    ab = lambda x, func: ansatz_batch(func, x, 100000, sorb, torch.device("cpu"), torch.double)  # noqa: E731
    x = I["x"][:16].contiguous()
    out = dict(x=x.numpy())
    for tag, Lambda in (("a", -100.0), ("b", -106.0)):  # b: Lambda - H_00 changes sign among the walkers -> the clamp branch
        with torch.no_grad():
            eloc, gk, comb_x, stop, mask = GFMC._calculate_green_kernel(None, x, Lambda, I["h1e"], I["h2e"], rbm, ab, sorb, nele, noA, noB,
                                                                        torch.double, None, True)
        g = torch.Generator().manual_seed(71)
        rand = torch.rand(16, 1, generator=g, dtype=torch.float64)
        rand[0, 0] = 0.0; rand[1, 0] = 0.999
        wgt = torch.rand(16, generator=g, dtype=torch.float64) + 0.5
        x_new, w_new, beta, acc = GFMC.sample_update(None, x, wgt, comb_x, gk, rand)
        out.update({f"{tag}_Lambda": Lambda, f"{tag}_eloc": eloc.numpy(), f"{tag}_gk4": gk[:4].numpy(), f"{tag}_gk_rowsum": gk.sum(-1).numpy(),
                    f"{tag}_gk_col0": gk[:, 0].numpy(), f"{tag}_mask": mask.numpy(), f"{tag}_rand": rand.numpy(), f"{tag}_weight": wgt.numpy(),
                    f"{tag}_x_new": x_new.numpy(), f"{tag}_w_new": w_new.numpy(), f"{tag}_beta": beta.numpy(), f"{tag}_accept": acc})
    for k, v in branch.items():
        out["branch_" + k] = v
    np.savez_compressed(f"{out_dir}/gfmc_fe2s2.npz", **out)
    print("gfmc done; accept:", out["a_accept"], out["b_accept"], "clamped:", int(out["a_mask"].sum()), int(out["b_mask"].sum()))


def section_helpers(mod, out_dir):
    """The two ansatz-side helpers the drop-in also provides: permute_sgn (cpu_tensor.cpp:356, onstate.cpp:195) and
    constrain_make_charts (cpu_tensor.cpp:558), run on the reference's compiled extension."""
    g = torch.Generator().manual_seed(91)
    out = {}
    for sorb in (8, 40):
        perms = torch.stack([torch.randperm(sorb, generator=g) for _ in range(6)] + [torch.arange(sorb)])
        occ = (torch.rand(50, sorb, generator=g) < 0.5).long()
        out[f"perm_{sorb}"], out[f"occ_{sorb}"] = perms.numpy(), occ.numpy()
        out[f"sgn_{sorb}"] = np.stack([mod.permute_sgn(p.contiguous(), occ, sorb).numpy() for p in perms])
    idx = torch.tensor([10, 6, 14, 9, 5, 13, 11, 7, 15, 15, 5, 10], dtype=torch.int64)
    out["chart_idx"], out["charts"] = idx.numpy(), mod.constrain_make_charts(idx).numpy()
    np.savez_compressed(f"{out_dir}/ansatz_helpers.npz", **out)
    print("helpers done:", out["sgn_40"].shape, out["charts"].shape)


def section_merge(mod, out_dir, samp, stats):
    out = dict(samp)
    # merge_rank_sample on its own: three "ranks" with overlapping determinants
    g = torch.Generator().manual_seed(81)
    inv = torch.randint(0, 40, (150,), generator=g)
    cnt = torch.randint(1, 1000, (150,), generator=g)
    split = torch.tensor([0, 60, 110, 150])
    out["mrs_inv"], out["mrs_counts"], out["mrs_split"], out["mrs_length"] = inv.numpy(), cnt.numpy(), split.numpy(), 40
    out["mrs_out"] = mod.merge_rank_sample(inv, cnt, split, 40).numpy()
    for k, v in stats.items():
        out["stats_" + k] = v
    np.savez_compressed(f"{out_dir}/sampler_merge.npz", **out)
    print("sampler merge done")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/refbuild")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_default_dtype(torch.float64)
    mod = harness(a.scratch)
    I = load_inputs()
    only = set(a.only.split(",")) if a.only else {"eloc", "dist"}
    if "eloc" in only:
        section_eloc(I, a.out)
    if "eloc" in only or "flavours" in only:
        section_rbm_flavours(I, a.out)
    if "eloc" in only or "spin" in only:
        section_spin_raising(I, a.out)
    if "eloc" in only or "helpers" in only:
        section_helpers(mod, a.out)
    if "dist" in only:
        branch, samp, stats = section_dist(a.scratch, a.out)
        section_gfmc(I, a.out, branch)
        section_merge(mod, a.out, samp, stats)


if __name__ == "__main__":
    main()

"""Spin-flip-projected, multi-psi and complex128-module local energies on the GPU against outputs captured from the
reference's own Python (vmc/energy/flip.py:66-418, the use_multi_psi branches of vmc/energy/eloc.py:134-401;
tests/golden/make_golden_r2.py -> eloc_flip_multipsi_fe2s2.npz, eloc_complex_module.npz).
Tolerance: 1e-8 Ha per determinant; psi(x) to 1e-12 relative."""
import types

import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
TOL = 1e-8
SYS = (40, 30, 15, 15)


class Holder:
    """ansatz.module.sample / ansatz.module.extra, as the reference's multi-psi code dereferences a DDP-wrapped model."""

    def __init__(self, sample, extra):
        self.module = types.SimpleNamespace(sample=sample, extra=extra)


@pytest.fixture(scope="module")
def env(fe2s2):
    from pynqs_amd import energy, public_function as pf
    from pynqs_amd.rbm import ComplexRBM, RealRBM

    assert torch.cuda.is_available()
    d0 = golden("eloc_e2e_fe2s2.npz")
    d = golden("eloc_flip_multipsi_fe2s2.npz")
    c = golden("eloc_complex_module.npz")
    dev = torch.device("cuda")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    torch.set_default_dtype(torch.float64)
    rbm = RealRBM(T(d0["W"]), T(d0["hb"]), T(d0["vb"])).to(dev)
    extra = RealRBM(T(d["W2"]), T(d["hb2"]), T(d["vb2"])).to(dev)
    crbm = ComplexRBM(T(d["Wc"]), T(d["hbc"]), T(d["vbc"])).to(dev)
    pf.SpinProjection.init(30, 0)
    assert pf.SpinProjection.eta == int(d["eta"])
    keys, wf, wfc = T(d["lut_keys"]), T(d["lut_wf"]), T(d["lut_wfc"])
    ns = int(d["n_lut_small"])
    e = dict(energy=energy, pf=pf, d=d, c=c, dev=dev, T=T, h1e=T(fe2s2["h1e"]), h2e=T(fe2s2["h2e"]), x=T(d["x"]), rbm=rbm, crbm=crbm,
             multi=Holder(rbm, extra), cmulti=Holder(rbm, crbm),
             en=torch.tensor(float(d["extra_norm"]), dtype=torch.float64, device=dev),
             enm=torch.tensor(float(d["extra_norm_multi"]), dtype=torch.float64, device=dev),
             lut=pf.WavefunctionLUT(keys, wf, 40, device=dev), lutc=pf.WavefunctionLUT(keys, wfc, 40, device=dev),
             lut_small=pf.WavefunctionLUT(keys[:ns].contiguous(), wf[:ns].contiguous(), 40, device=dev),
             lutk=pf.WavefunctionLUT(T(c["lut_keys"]), T(c["lut_wf"]), 40, device=dev))
    yield e
    torch.set_default_dtype(torch.float32)


def _le(env, ansatz, dt=torch.double, fused=True, **kw):
    energy, pf = env["energy"], env["pf"]
    ab = lambda x, func: pf.ansatz_batch(func, x, 100000, 40, env["dev"], dt)  # noqa: E731
    old = energy.FUSED
    energy.FUSED = fused
    try:
        e, s, p, _ = energy.local_energy(env["x"], env["h1e"], env["h2e"], ansatz, ab, *SYS, dtype=dt, **kw)
    finally:
        energy.FUSED = old
    assert e.dtype == dt and p.dtype == dt and float(s.abs().max()) == 0.0
    return e.cpu().numpy(), p.cpu().numpy()


def _check(got, d, name):
    e, p = got
    np.testing.assert_allclose(p, d["psi_" + name], rtol=1e-12)
    np.testing.assert_allclose(e, d["eloc_" + name], rtol=0, atol=TOL)


CASES = {
    # name: (ansatz key, dtype, kwargs builder)
    "simple_flip": ("rbm", torch.double, lambda v: dict(use_spin_flip=True, extra_norm=v["en"])),
    "reduce_flip": ("rbm", torch.double, lambda v: dict(use_spin_flip=True, extra_norm=v["en"], reduce_psi=True, eps=1e-2, eps_sample=0)),
    "reduce_flip_lut": ("rbm", torch.double, lambda v: dict(use_spin_flip=True, extra_norm=v["en"], reduce_psi=True, eps=1e-2, eps_sample=0,
                                                             WF_LUT=v["lut_small"])),
    "ss_flip": ("rbm", torch.double, lambda v: dict(use_spin_flip=True, extra_norm=v["en"], use_sample_space=True, WF_LUT=v["lut"], index=(0, 32))),
    "ss_flip_c": ("rbm", torch.complex128, lambda v: dict(use_spin_flip=True, extra_norm=v["en"], use_sample_space=True, WF_LUT=v["lutc"],
                                                          index=(0, 32))),
    "simple_multi": ("multi", torch.double, lambda v: dict(use_multi_psi=True, extra_norm=v["enm"])),
    "reduce_multi": ("multi", torch.double, lambda v: dict(use_multi_psi=True, extra_norm=v["enm"], reduce_psi=True, eps=1e-2, eps_sample=0)),
    "ss_multi": ("multi", torch.double, lambda v: dict(use_multi_psi=True, extra_norm=v["enm"], use_sample_space=True, WF_LUT=v["lut"], index=(0, 32))),
    "simple_multi_c": ("cmulti", torch.complex128, lambda v: dict(use_multi_psi=True, extra_norm=v["enm"])),
    "simple_flip_multi": ("multi", torch.double, lambda v: dict(use_spin_flip=True, use_multi_psi=True, extra_norm=v["enm"])),
    "reduce_flip_multi": ("multi", torch.double, lambda v: dict(use_spin_flip=True, use_multi_psi=True, extra_norm=v["enm"], reduce_psi=True,
                                                                eps=1e-2, eps_sample=0)),
    "ss_flip_multi": ("multi", torch.double, lambda v: dict(use_spin_flip=True, use_multi_psi=True, extra_norm=v["enm"], use_sample_space=True,
                                                            WF_LUT=v["lut"], index=(0, 32))),
    "ss_flip_multi_c": ("cmulti", torch.complex128, lambda v: dict(use_spin_flip=True, use_multi_psi=True, extra_norm=v["enm"],
                                                                   use_sample_space=True, WF_LUT=v["lutc"], index=(0, 32))),
}


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", sorted(CASES))
def test_projected_and_multi_psi_match_reference_python(env, name, fused):
    key, dt, kw = CASES[name]
    _check(_le(env, env[key], dt, fused, **kw(env)), env["d"], name)


@pytest.mark.parametrize("fused", [True, False])
def test_complex128_module_matches_reference_python(env, fused):
    """C4's amplitude dtype on the generic path: SIMPLE, REDUCE (on-chip compaction when fused), REDUCE + LUT, SIMPLE + spin flip
    with a complex-valued nn.Module."""
    c, cr = env["c"], env["crbm"]
    _check(_le(env, cr, torch.complex128, fused), c, "simple")
    _check(_le(env, cr, torch.complex128, fused, reduce_psi=True, eps=1e-2, eps_sample=0), c, "reduce")
    e, _ = _le(env, cr, torch.complex128, fused, reduce_psi=True, eps=1e-2, eps_sample=0, WF_LUT=env["lutk"])
    np.testing.assert_allclose(e, c["eloc_reduce_lut"], rtol=0, atol=TOL)
    _check(_le(env, cr, torch.complex128, fused, use_spin_flip=True, extra_norm=env["en"]), c, "simple_flip")
    # no use_unique: same numbers (Func without the unique pass)
    e2, _ = _le(env, cr, torch.complex128, fused, use_unique=False)
    np.testing.assert_allclose(e2, c["eloc_simple"], rtol=0, atol=TOL)


def test_projected_sample_space_runs_fused(env, monkeypatch):
    """The SAMPLE_SPACE forms with use_spin_flip / use_multi_psi take the one-kernel path (pynqs_eloc_sample_space[_hash][_flip]):
    nothing of size batch x ncomb may be materialised."""
    energy = env["energy"]

    def boom(*a, **k):
        raise AssertionError("get_comb_hij_fused called: the projected SAMPLE_SPACE form fell back to the materialising path")

    monkeypatch.setattr(energy, "get_comb_hij_fused", boom)
    for name in ("ss_flip", "ss_flip_c", "ss_multi", "ss_flip_multi", "ss_flip_multi_c"):
        key, dt, kw = CASES[name]
        _check(_le(env, env[key], dt, True, **kw(env)), env["d"], name)


def test_projected_reduce_runs_compacted(env, monkeypatch):
    """REDUCE with use_spin_flip / use_multi_psi / a complex module keeps the on-chip compaction (pynqs_reduce_count / _emit): the
    extra factors f, psi(flip x'), eta_m are evaluated on the kept records, nothing of size batch x ncomb is materialised."""
    energy = env["energy"]

    def boom(*a, **k):
        raise AssertionError("get_comb_hij_fused called: the projected REDUCE form fell back to the materialising path")

    monkeypatch.setattr(energy, "get_comb_hij_fused", boom)
    for name in ("reduce_flip", "reduce_flip_lut", "reduce_multi", "reduce_flip_multi"):
        key, dt, kw = CASES[name]
        _check(_le(env, env[key], dt, True, **kw(env)), env["d"], name)


@pytest.mark.parametrize("sorb,no,nkeys,use_hash", [(40, 5, 300, True), (72, 6, 200, True), (136, 4, 3000, True), (136, 4, 5000, True),
                                                    (72, 6, 300, False), (40, 5, 300_000, True)])
def test_spin_flip_kernel_all_filter_levels(sorb, no, nkeys, use_hash):
    """The partner sum of the projected form (pynqs_eloc_sample_space[_hash]_flip) with one-level / two-level / no filter, sorted-key
    search, 1-3 ONV words, against the materialising tensor path (the reference's algebra, pinned at sorb 40 by the fixtures above)
    on sample spaces closed under the alpha <-> beta exchange."""
    from pynqs_amd import energy, public_function as pf
    from test_gpu_energy import _excite, _random_walkers

    rng = np.random.default_rng(7000 * sorb + nkeys)
    dev = torch.device("cuda")
    n = 16
    x = _random_walkers(rng, n, sorb, no)
    pool = np.concatenate([x, _excite(rng, x, sorb, nkeys // 3, True), _excite(rng, x, sorb, nkeys // 3, False), _random_walkers(rng, nkeys, sorb, no)])
    L = x.shape[1]
    tb = lambda w: torch.from_numpy(np.ascontiguousarray(w).view(np.uint8).reshape(-1, 8 * L)).to(dev)  # noqa: E731
    keys = torch.unique(torch.cat([tb(pool), pf.spin_flip_onv(tb(pool), sorb)]), dim=0)
    wf = torch.from_numpy(rng.standard_normal(keys.size(0)) + 1j * rng.standard_normal(keys.size(0))).to(dev)
    h1 = rng.standard_normal((sorb, sorb)); h1 = torch.from_numpy((h1 + h1.T).reshape(-1)).to(dev)
    pair = sorb * (sorb - 1) // 2
    h2 = torch.from_numpy(rng.standard_normal(pair * (pair + 1) // 2)).to(dev)
    old_flag = pf.USE_HASH
    pf.USE_HASH = use_hash
    try:
        lut = pf.WavefunctionLUT(keys, wf, sorb, device=dev)
    finally:
        pf.USE_HASH = old_flag
    pf.SpinProjection.init(2 * no, 0)
    en = torch.tensor(1.7, dtype=torch.float64, device=dev)
    out = {}
    for fused in (True, False):
        old = energy.FUSED
        energy.FUSED = fused
        try:
            e, _, p0, _ = energy.local_energy(tb(x), h1, h2, None, lambda x_, func: None, sorb, 2 * no, no, no, WF_LUT=lut, use_sample_space=True,
                                              dtype=torch.complex128, use_spin_flip=True, extra_norm=en)
        finally:
            energy.FUSED = old
        out[fused] = (e.cpu().numpy(), p0.cpu().numpy())
    np.testing.assert_array_equal(out[True][1], out[False][1])
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=0, atol=TOL, err_msg=f"|E_loc|max = {float(np.abs(out[False][0]).max()):.6g} Ha")
    pf.SpinProjection.init(30, 0)


@pytest.mark.parametrize("fused", [True, False])
def test_spin_raising_matches_reference_python(env, fused):
    """use_spin_raising: <S-S+> next to the energy (eloc.py:173-188,250-310,377-400) with the reference's own S-S+ integrals
    (utils/pyscf_helper/operator.py:93-137, part of the fixture), for SIMPLE (fused: the RBM kernel run with both integral sets),
    REDUCE, SAMPLE_SPACE (fused: a second pass of the one-kernel form) and total_energy's REDUCE + sample-space form (etot.py:93-142)."""
    energy, pf, T, dev = env["energy"], env["pf"], env["T"], env["dev"]
    s = golden("eloc_spin_raising_fe2s2.npz")
    h1s, h2s = T(s["h1e_spin"]), T(s["h2e_spin"])
    lut = pf.WavefunctionLUT(T(s["lut_keys"]), T(s["lut_wf"]), 40, device=dev)
    ab = lambda x, func: pf.ansatz_batch(func, x, 100000, 40, dev, torch.double)  # noqa: E731
    old = energy.FUSED, energy.FUSED_RBM
    energy.FUSED = energy.FUSED_RBM = fused
    try:
        for tag, kw in (("simple", {}), ("reduce", dict(reduce_psi=True, eps=1e-2, eps_sample=0)),
                        ("ss", dict(use_sample_space=True, WF_LUT=lut, index=(0, 32)))):
            e, sl, p, _ = energy.local_energy(env["x"], env["h1e"], env["h2e"], env["rbm"], ab, *SYS, use_spin_raising=True, h1e_spin=h1s, h2e_spin=h2s,
                                              **kw)
            np.testing.assert_allclose(e.cpu().numpy(), s["eloc_" + tag], rtol=0, atol=TOL)
            np.testing.assert_allclose(sl.cpu().numpy(), s["sloc_" + tag], rtol=0, atol=TOL)
            np.testing.assert_allclose(p.cpu().numpy(), s["psi_" + tag], rtol=1e-12)
        e, sl, _ = energy.total_energy(env["x"], 16, 100000, env["h1e"], env["h2e"], env["rbm"], *SYS, WF_LUT=lut, use_spin_raising=True,
                                       h1e_spin=h1s, h2e_spin=h2s, reduce_psi=True, eps=1e-2, eps_sample=0)
        np.testing.assert_allclose(e.cpu().numpy(), s["eloc_etot_reduce"], rtol=0, atol=TOL)
        np.testing.assert_allclose(sl.cpu().numpy(), s["sloc_etot_reduce"], rtol=0, atol=TOL)
    finally:
        energy.FUSED, energy.FUSED_RBM = old


@pytest.mark.parametrize("key_major", [True, False, "indexed"])
def test_sample_space_kernels_key_major_and_column_major(env, key_major):
    """The fused SAMPLE_SPACE kernels -- walking the table (pynqs_eloc_sample_space_keys: every key; pynqs_eloc_sample_space_indexed: the
    keys that share a block of orbitals with the walker) or the excitation lists (pynqs_eloc_sample_space[_hash][_flip]) -- forced in
    turn (the energy layer otherwise picks by table size against ncomb and by the index's density):
    every projected / multi-psi / complex sample-space fixture, <S-S+>, and random 1-3-word systems with and without hits."""
    energy, pf, T, dev = env["energy"], env["pf"], env["T"], env["dev"]
    old = energy.SS_KEYS, energy.SS_INDEX
    energy.SS_KEYS, energy.SS_INDEX = bool(key_major), key_major == "indexed"
    try:
        for name in ("ss_flip", "ss_flip_c", "ss_multi", "ss_flip_multi", "ss_flip_multi_c"):
            key, dt, kw = CASES[name]
            _check(_le(env, env[key], dt, True, **kw(env)), env["d"], name)
        s = golden("eloc_spin_raising_fe2s2.npz")
        lut = pf.WavefunctionLUT(T(s["lut_keys"]), T(s["lut_wf"]), 40, device=dev)
        ab = lambda x, func: pf.ansatz_batch(func, x, 100000, 40, dev, torch.double)  # noqa: E731
        e, sl, p, _ = energy.local_energy(env["x"], env["h1e"], env["h2e"], env["rbm"], ab, *SYS, use_spin_raising=True, h1e_spin=T(s["h1e_spin"]),
                                          h2e_spin=T(s["h2e_spin"]), use_sample_space=True, WF_LUT=lut, index=(0, 32))
        np.testing.assert_allclose(e.cpu().numpy(), s["eloc_ss"], rtol=0, atol=TOL)
        np.testing.assert_allclose(sl.cpu().numpy(), s["sloc_ss"], rtol=0, atol=TOL)
        np.testing.assert_allclose(p.cpu().numpy(), s["psi_ss"], rtol=1e-12)
        for args in ((40, 5, 300, True), (72, 6, 200, True), (136, 4, 3000, True), (72, 6, 300, False)):
            test_spin_flip_kernel_all_filter_levels(*args)
    finally:
        energy.SS_KEYS, energy.SS_INDEX = old


def test_key_major_kernel_edge_cases(env):
    """pynqs_eloc_sample_space_keys: unsorted keys, walkers that are not in the table (psi(x) = 0: E_loc is inf / nan as in the
    reference's division), a number of walkers that does not fill the last wave, a table of one key."""
    from pynqs_amd import C_extension as cx, _native as N

    dev = env["dev"]
    x = env["x"][:13].contiguous()
    keys = torch.cat([env["x"][5:32], env["x"][:3]]).contiguous()  # unsorted; walkers 3, 4 are missing
    wf = torch.rand(keys.size(0), dtype=torch.float64, device=dev) + 0.5
    plan = cx.plan_for(env["h1e"], env["h2e"], 40, dev)
    e = torch.empty(13, dtype=torch.float64, device=dev); p0 = torch.empty(13, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    N.check(N.lib().pynqs_eloc_sample_space_keys(x.data_ptr(), 13, 40, 30, 15, 15, plan.data_ptr(), keys.data_ptr(), keys.size(0), wf.data_ptr(), 0, 0,
                                                 e.data_ptr(), p0.data_ptr(), st), "keys")
    comb, hm = cx.get_comb_hij_fused(x, env["h1e"], env["h2e"], 40, 30, 15, 15)
    table = {bytes(k.cpu().numpy().tobytes()): float(v) for k, v in zip(keys, wf)}
    psi = torch.tensor([[table.get(bytes(c.cpu().numpy().tobytes()), 0.0) for c in row] for row in comb[:, :, :8]], dtype=torch.float64, device=dev)
    want0 = psi[:, 0]
    assert torch.equal(p0, want0) and float(p0[3]) == 0.0 and float(p0[4]) == 0.0
    with np.errstate(all="ignore"):
        want = ((hm * psi).sum(1) / want0).cpu().numpy()
    got = e.cpu().numpy()
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=0, atol=TOL)
    assert not np.isfinite(got[~ok]).any() and (~ok).sum() == 2
    one = keys[:1].contiguous()
    N.check(N.lib().pynqs_eloc_sample_space_keys(x.data_ptr(), 13, 40, 30, 15, 15, plan.data_ptr(), one.data_ptr(), 1, wf.data_ptr(), 0, 0,
                                                 e.data_ptr(), p0.data_ptr(), st), "keys")
    assert int((p0 != 0).sum()) == 1  # only the walker equal to that key (x[5]) has psi(x) != 0


def test_sample_space_kernel_choice_is_probed_between_the_clear_cases(env, fe2s2, monkeypatch):
    """A table of 2.3 x ncomb keys (Fe2S2's whole CI space) is neither clearly small nor clearly large.  By default the fixed ratio decides
    (the same kernel on every rank and in every run, no timing inside a step); with SS_AUTOTUNE (PYNQS_SS_AUTOTUNE=1) the first call times
    both kernels on the walkers at hand and remembers the winner for that (system, table-size bucket).  The energies match the oracle
    whichever kernel runs."""
    from oracle import oracle as O

    energy, pf, T, dev = env["energy"], env["pf"], env["T"], env["dev"]
    keys = T(fe2s2["ci_space"])
    g = torch.Generator().manual_seed(3)
    wf = (torch.rand(keys.size(0), generator=g, dtype=torch.float64) + 0.2).to(dev)
    lut = pf.WavefunctionLUT(keys, wf, 40, device=dev)
    monkeypatch.delenv("PYNQS_SS_KEYS", raising=False)
    assert energy.SS_KEYS is None and not energy.SS_AUTOTUNE  # (deterministic by default)
    energy._SS_CHOICE.clear()
    x = T(fe2s2["ci_space"][:96])
    e0, _, _, _ = energy.local_energy(x, env["h1e"], env["h2e"], None, None, *SYS, WF_LUT=lut, use_sample_space=True)
    assert len(energy._SS_CHOICE) == 0  # nothing was timed
    monkeypatch.setattr(energy, "SS_AUTOTUNE", True)
    e, _, p0, _ = energy.local_energy(x, env["h1e"], env["h2e"], None, None, *SYS, WF_LUT=lut, use_sample_space=True)
    assert len(energy._SS_CHOICE) == 1 and isinstance(next(iter(energy._SS_CHOICE.values())), bool)
    e2, _, _, _ = energy.local_energy(x, env["h1e"], env["h2e"], None, None, *SYS, WF_LUT=lut, use_sample_space=True)
    assert len(energy._SS_CHOICE) == 1  # remembered
    e_ref, p_ref = O.eloc_sample_space(fe2s2["ci_space"][:96].copy(), fe2s2["h1e"], fe2s2["h2e"], 40, 30, 15, 15, lut.bra_key.cpu().numpy(),
                                       lut.wf_value.cpu().numpy())
    np.testing.assert_array_equal(p0.cpu().numpy(), p_ref)
    np.testing.assert_allclose(e.cpu().numpy(), e_ref, rtol=0, atol=TOL)
    np.testing.assert_allclose(e2.cpu().numpy(), e_ref, rtol=0, atol=TOL)
    np.testing.assert_allclose(e0.cpu().numpy(), e_ref, rtol=0, atol=TOL)
    energy._SS_CHOICE.clear()


def test_fused_sample_space_accepts_the_reference_lut_class_shape(env):
    """With the one-line change of INTEGRATION.md the tables handed to local_energy are the REFERENCE's WavefunctionLUT objects: sorted
    keys, values, `sort`, `dtype` -- no hash table (USE_HASH = False there) and no `find`.  A stand-in with exactly those attributes
    must take the fused kernels, all variants."""
    energy, d = env["energy"], env["d"]

    class RefShapedLUT:  # the attribute surface of utils/public_function.py:749-868
        def __init__(self, lut):
            self.sort, self.sorb = True, lut.sorb
            self._bra_key, self._wf_value = lut.bra_key, lut.wf_value

        bra_key = property(lambda self: self._bra_key)
        wf_value = property(lambda self: self._wf_value)
        dtype = property(lambda self: self._wf_value.dtype)

        def lookup(self, onv):
            raise AssertionError("the fused path must not call lookup()")

    for name in ("ss_flip", "ss_multi", "ss_flip_multi_c"):
        key, dt, kw = CASES[name]
        k = kw(env)
        k["WF_LUT"] = RefShapedLUT(k["WF_LUT"])
        for mode in (True, False):
            old = energy.SS_KEYS
            energy.SS_KEYS = mode
            try:
                _check(_le(env, env[key], dt, True, **k), d, name)
            finally:
                energy.SS_KEYS = old

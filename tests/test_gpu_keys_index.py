"""INDEXED key-major SAMPLE_SPACE local energy (pynqs_keys_index_build / pynqs_eloc_sample_space_indexed; vmc/energy/eloc.py:326-401):
the block index itself against a numpy restatement of its definition, the kernel against the oracle and the streamed / column-major
kernels (1e-8 Ha absolute; psi(x) bit-exact), edge cases, bit-reproducibility, and the streamed-or-indexed rule of the energy layer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-8
NB = 5


def _block_values(k64, sorb, b):
    """bits [lo_b, lo_{b+1}) of every key; lo_b = 2 * ((b * (sorb / 2)) / 5) (include/pynqs_amd.h, detcore.h: index_block_lo)"""
    L = k64.shape[1]
    lo, hi = 2 * ((b * (sorb // 2)) // NB), 2 * (((b + 1) * (sorb // 2)) // NB)
    w, sh = lo >> 6, lo & 63
    v = k64[:, w] >> np.uint64(sh)
    if sh and w + 1 < L:
        v = v | (k64[:, w + 1] << np.uint64(64 - sh))
    return v & np.uint64((1 << (hi - lo)) - 1), hi - lo


def _synthetic(sorb, no, n, nkeys, seed):
    import bench as B

    x = B.synth_walkers(n, sorb, no, no, seed)
    more = B.synth_connected(x, sorb, nkeys - n, seed + 1)
    keys = torch.unique(torch.cat([x, more]), dim=0)
    h1, h2 = B.synth_integrals(sorb)
    return x, keys[torch.randperm(keys.size(0), generator=torch.Generator().manual_seed(seed))].contiguous(), h1, h2  # (any order)


@pytest.mark.parametrize("sorb,no", [(8, 2), (40, 15), (56, 7), (120, 30), (184, 46), (192, 20)])
def test_index_is_the_keys_sorted_by_each_block(sorb, no):
    from pynqs_amd import C_extension as cx

    dev = torch.device("cuda")
    _, keys, _, _ = _synthetic(sorb, no, 64, 3000, 11)
    ki = cx.keys_index_build(keys.to(dev), sorb)
    nk, L = keys.size(0), keys.size(1) // 8
    raw = ki.index.cpu().numpy().view(np.uint8)
    svals = raw[: NB * nk * 8].view(np.uint64).reshape(NB, nk)
    perm = raw[NB * nk * 8: NB * nk * 12].view(np.uint32).reshape(NB, nk)
    k64 = keys.numpy().view(np.uint64).reshape(nk, L)
    total = 0
    for b in range(NB):
        v, width = _block_values(k64, sorb, b)
        assert width <= 40
        order = np.argsort(v, kind="stable")  # equal values keep the order of the key array
        np.testing.assert_array_equal(perm[b], order.astype(np.uint32))
        np.testing.assert_array_equal(svals[b], v[order] | np.uint64(b << 40))
        total += int((np.unique(v, return_counts=True)[1].astype(np.int64) ** 2).sum())
    assert ki.per_walker == total / nk
    assert ki.memory >= 60 * nk


@pytest.mark.parametrize("sorb,no,cplx", [(8, 2, False), (40, 15, True), (56, 7, False), (120, 30, True), (184, 46, True), (184, 46, False)])
def test_indexed_kernel_against_oracle_and_streamed(sorb, no, cplx):
    from oracle import oracle as O
    from pynqs_amd import C_extension as cx, _native as N

    from pynqs_amd import public_function as pf

    dev = torch.device("cuda")
    n, nkeys = (77, 4000) if sorb > 8 else (20, 30)  # (77: the last workgroup is not full)
    x, keys, h1, h2 = _synthetic(sorb, no, n, nkeys, 5)
    nk = keys.size(0)
    g = torch.Generator().manual_seed(9)
    wf = torch.rand(nk, generator=g, dtype=torch.float64) + 0.25
    if cplx:
        wf = torch.polar(wf, 6.28 * torch.rand(nk, generator=g, dtype=torch.float64))
    xd, kd, wd = x.to(dev), keys.to(dev), wf.to(dev)
    plan = cx.plan_for(h1.to(dev), h2.to(dev), sorb, dev)
    ki = cx.keys_index_build(kd, sorb)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for name in ("streamed", "indexed", "indexed again"):
        e, p0 = torch.full_like(wd[:n], 7.0), torch.full_like(wd[:n], 7.0)
        args = (xd.data_ptr(), n, sorb, 2 * no, no, no, plan.data_ptr(), kd.data_ptr(), nk)
        if name == "streamed":
            rc = N.lib().pynqs_eloc_sample_space_keys(*args, wd.data_ptr(), int(cplx), 0, e.data_ptr(), p0.data_ptr(), st)
        else:
            rc = N.lib().pynqs_eloc_sample_space_indexed(*args, ki.index.data_ptr(), wd.data_ptr(), int(cplx), 0, e.data_ptr(), p0.data_ptr(), st)
        N.check(rc, name)
        out[name] = (e.cpu().numpy(), p0.cpu().numpy())
    lut = pf.WavefunctionLUT(keys, wf, sorb, device="cpu")  # (the oracle searches a sorted table; the kernels got the keys shuffled)
    e_ref, p_ref = O.eloc_sample_space(x.numpy().copy(), h1.numpy(), h2.numpy(), sorb, 2 * no, no, no, lut.bra_key.numpy().copy(), lut.wf_value.numpy())
    for name in ("streamed", "indexed"):
        np.testing.assert_array_equal(out[name][1], p_ref, err_msg=name)
        np.testing.assert_allclose(out[name][0], e_ref, rtol=0, atol=TOL, err_msg=f"{name}: |E_loc|max = {float(np.abs(e_ref).max()):.6g} Ha")
    np.testing.assert_array_equal(out["indexed"][0], out["indexed again"][0])  # no atomics: the same bits every time


def test_indexed_kernel_edge_cases(fe2s2):
    """Walkers that are not in the table (psi(x) = 0: inf / nan as in the reference's division), a table of one key, no keys at all, bad
    arguments."""
    from pynqs_amd import C_extension as cx, _native as N

    dev = torch.device("cuda")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ci = T(fe2s2["ci_space"][:64])
    h1e, h2e = T(fe2s2["h1e"]), T(fe2s2["h2e"])
    x = ci[:13].contiguous()
    keys = torch.cat([ci[5:32], ci[:3]]).contiguous()  # walkers 3, 4 are missing
    wf = torch.rand(keys.size(0), dtype=torch.float64, device=dev) + 0.5
    plan = cx.plan_for(h1e, h2e, 40, dev)
    st = torch.cuda.current_stream().cuda_stream
    lib = N.lib()

    def run(kk, indexed):
        e, p0 = torch.empty(13, dtype=torch.float64, device=dev), torch.empty(13, dtype=torch.float64, device=dev)
        args = (x.data_ptr(), 13, 40, 30, 15, 15, plan.data_ptr(), kk.data_ptr(), kk.size(0))
        if indexed:
            ki = cx.keys_index_build(kk, 40)
            N.check(lib.pynqs_eloc_sample_space_indexed(*args, ki.index.data_ptr(), wf.data_ptr(), 0, 0, e.data_ptr(), p0.data_ptr(), st), "indexed")
        else:
            N.check(lib.pynqs_eloc_sample_space_keys(*args, wf.data_ptr(), 0, 0, e.data_ptr(), p0.data_ptr(), st), "keys")
        return e.cpu().numpy(), p0.cpu().numpy()

    for kk in (keys, keys[:1].contiguous()):
        (es, ps), (ei, pi) = run(kk, False), run(kk, True)
        np.testing.assert_array_equal(ps, pi)
        ok = ps != 0
        assert (~ok).sum() == (2 if kk.size(0) > 1 else 12)
        np.testing.assert_allclose(ei[ok], es[ok], rtol=0, atol=TOL)
        assert not np.isfinite(ei[~ok]).any()
    # no keys: psi(x) = 0 everywhere, 0 / 0
    e, p0 = torch.empty(13, dtype=torch.float64, device=dev), torch.ones(13, dtype=torch.float64, device=dev)
    N.check(lib.pynqs_eloc_sample_space_indexed(x.data_ptr(), 13, 40, 30, 15, 15, plan.data_ptr(), 0, 0, 0, 0, 0, 0, e.data_ptr(), p0.data_ptr(), st), "empty")
    assert float(p0.abs().max()) == 0.0 and bool(torch.isnan(e).all())
    assert lib.pynqs_keys_index_bytes(10, 41) == -1 and lib.pynqs_keys_index_bytes(1 << 27, 40) == -1 and lib.pynqs_keys_index_bytes(0, 40) == 0
    assert lib.pynqs_eloc_sample_space_indexed(x.data_ptr(), 13, 40, 30, 15, 15, plan.data_ptr(), keys.data_ptr(), keys.size(0), 0, wf.data_ptr(), 0, 0,
                                               e.data_ptr(), p0.data_ptr(), st) == -1  # PYNQS_EINVAL
    assert b"index" in lib.pynqs_last_error()


def test_energy_layer_buys_the_index_when_streaming_has_paid_for_it(fe2s2, monkeypatch):
    """local_energy, key-major: a small call streams (no index, the pairs are counted on the table); once the pairs streamed against
    the table would have paid for the index it is built (once) and used -- unless the table is dense (Fe2S2's CI space: a member shares
    its blocks with more keys than the table has), where the streamed form stays.  Same energies either way."""
    from pynqs_amd import energy, public_function as pf

    dev = torch.device("cuda")
    monkeypatch.delenv("PYNQS_SS_INDEX", raising=False)
    monkeypatch.setattr(energy, "SS_KEYS", True)
    monkeypatch.setattr(energy, "SS_INDEX", None)
    sorb, no = 120, 30
    x, keys, h1, h2 = _synthetic(sorb, no, 4096, 32768, 21)
    wf = torch.rand(keys.size(0), dtype=torch.float64, generator=torch.Generator().manual_seed(1)) + 0.3
    lut = pf.WavefunctionLUT(keys.to(dev), wf.to(dev), sorb, device=dev)
    h1, h2, xd = h1.to(dev), h2.to(dev), x.to(dev)
    run = lambda xs: energy.local_energy(xs, h1, h2, None, None, sorb, 2 * no, no, no, WF_LUT=lut, use_sample_space=True)[0]  # noqa: E731
    e_small = run(xd[:64])
    assert getattr(lut, "_keys_index", None) is None and lut._keys_streamed_pairs == 64 * lut.bra_key.size(0)
    e_all = run(xd)  # 4096 x 32768 pairs x 3.7e-13 s = 50 us < the index's ~90 us: still streamed
    assert getattr(lut, "_keys_index", None) is None
    e_all2 = run(xd)  # ... but now it has paid
    ki = lut._keys_index
    assert ki is not None and ki.per_walker < 64
    assert run(xd[:64]) is not None and lut._keys_index is ki  # built once
    torch.testing.assert_close(e_all2, e_all, rtol=0, atol=1e-10)
    torch.testing.assert_close(run(xd[:64]), e_small, rtol=0, atol=1e-10)
    # dense table: built, judged, not used
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    ci = T(fe2s2["ci_space"])
    lut2 = pf.WavefunctionLUT(ci, torch.rand(ci.size(0), dtype=torch.float64, device=dev) + 0.3, 40, device=dev)
    monkeypatch.setattr(energy, "SS_INDEX", True)
    e_idx = energy.local_energy(ci[:256].contiguous(), T(fe2s2["h1e"]), T(fe2s2["h2e"]), None, None, 40, 30, 15, 15, WF_LUT=lut2, use_sample_space=True)[0]
    assert lut2._keys_index.per_walker > ci.size(0)
    monkeypatch.setattr(energy, "SS_INDEX", None)
    assert energy._keys_index_for(lut2, 256, 40) is None
    e_str = energy.local_energy(ci[:256].contiguous(), T(fe2s2["h1e"]), T(fe2s2["h2e"]), None, None, 40, 30, 15, 15, WF_LUT=lut2, use_sample_space=True)[0]
    torch.testing.assert_close(e_idx, e_str, rtol=0, atol=1e-10)

"""HIP kernels (through the C ABI / pynqs_amd.C_extension) against the golden vectors captured from the
reference and against the CPU oracle on seeded inputs.  comb is bit-exact; Hmat is compared bit-exact
too (the kernels keep the reference's floating-point operation order), in f64 and f32."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import golden, rand_occ, synth_integrals

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module", params=["plan", "direct"])
def cx(request):
    """Both code paths of get_comb_hij_fused: the integral-plan kernels (default) and the direct
    packed-triangle kernels."""
    from pynqs_amd import C_extension as m
    from pynqs_amd import _native

    _native.lib()  # fail loudly if the HIP library is missing
    assert torch.cuda.is_available()
    old = m.USE_PLAN
    m.USE_PLAN = request.param == "plan"
    yield m
    m.USE_PLAN = old


def G(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_docstring_examples(cx):
    d = golden("docstring_examples.npz")
    torch.set_default_dtype(torch.float64)
    bra = torch.tensor([[0b1100, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.uint8).cuda()
    onv, states = cx.get_comb_tensor(bra, 4, 2, 1, 1, True)
    assert np.array_equal(onv.cpu().numpy(), d["comb_onv"]) and np.array_equal(states.cpu().numpy(), d["comb_states"])
    t = cx.tensor_to_onv(torch.tensor([1, 1, 1, 1, 0, 0, 0, 0], dtype=torch.uint8).cuda(), 8)
    assert np.array_equal(t.cpu().numpy(), d["t2o"])
    o = cx.onv_to_tensor(torch.tensor([[0b1111, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.uint8).cuda(), 8)
    assert np.array_equal(o.cpu().numpy(), d["o2t"])
    torch.set_default_dtype(torch.float32)


def test_c1_exhaustive(cx):
    d = golden("c1_sorb8_all36.npz")
    sorb, noA, noB = int(d["sorb"]), int(d["noA"]), int(d["noB"])
    nele = noA + noB
    onv = cx.tensor_to_onv(G(d["occ"]), sorb)
    assert np.array_equal(onv.cpu().numpy(), d["onv"])
    h1, h2 = G(d["h1e"]), G(d["h2e"])
    comb, hm = cx.get_comb_hij_fused(onv, h1, h2, sorb, nele, noA, noB)
    assert np.array_equal(comb.cpu().numpy(), d["comb"])
    assert np.array_equal(hm.cpu().numpy(), d["hmat"])
    comb32, hm32 = cx.get_comb_hij_fused(onv, h1.float(), h2.float(), sorb, nele, noA, noB)
    assert hm32.dtype == torch.float32 and np.array_equal(hm32.cpu().numpy(), d["hmat_f32"])
    c2, pm = cx.get_comb_tensor(onv, sorb, nele, noA, noB, True)
    assert np.array_equal(c2.cpu().numpy(), d["comb"]) and np.array_equal(pm.cpu().numpy(), d["comb_pm1"])
    c3, one = cx.get_comb_tensor(onv, sorb, nele, noA, noB)
    assert torch.equal(c3, c2) and one.device.type == "cpu" and one.dtype == torch.float64 and one.tolist() == [1.0]
    assert np.array_equal(cx.get_hij_torch(onv, comb, h1, h2, sorb, nele).cpu().numpy(), d["hmat"])
    assert np.array_equal(cx.get_hij_torch(onv, onv, h1, h2, sorb, nele).cpu().numpy(), d["hij2d"])
    assert np.array_equal(cx.get_hij_torch(onv, onv, h1.float(), h2.float(), sorb, nele).cpu().numpy(), d["hij2d_f32"])
    torch.set_default_dtype(torch.float64)
    assert np.array_equal(cx.onv_to_tensor(onv, sorb).cpu().numpy(), d["pm1"])
    torch.set_default_dtype(torch.float32)
    p32 = cx.onv_to_tensor(onv, sorb)
    assert p32.dtype == torch.float32 and np.array_equal(p32.cpu().numpy(), d["pm1_f32"])


def _cases(fname):
    d = golden(fname)
    return d, sorted({k.rsplit("_", 1)[0] for k in d.files if k.endswith("_onv")})


def test_asymmetric_small(cx):
    d, keys = _cases("asym_small.npz")
    for key in keys:
        sorb, noA, noB = (int(t[1:]) for t in key.split("_"))
        h1, h2 = synth_integrals(sorb)
        onv = G(d[key + "_onv"])
        comb, hm = cx.get_comb_hij_fused(onv, G(h1), G(h2), sorb, noA + noB, noA, noB)
        assert np.array_equal(comb.cpu().numpy(), d[key + "_comb"]), key
        assert np.array_equal(hm.cpu().numpy(), d[key + "_hmat"]), key
        _, hm32 = cx.get_comb_hij_fused(onv, G(h1).float(), G(h2).float(), sorb, noA + noB, noA, noB)
        assert np.array_equal(hm32.cpu().numpy(), d[key + "_hmat_f32"]), key
        assert np.array_equal(cx.get_hij_torch(onv, comb, G(h1), G(h2), sorb, noA + noB).cpu().numpy(), d[key + "_hmat"])


def test_word_boundaries(cx):
    d, keys = _cases("word_boundary.npz")
    torch.set_default_dtype(torch.float64)
    for key in keys:
        sorb, noA, noB = (int(t[1:]) for t in key.split("_"))
        h1, h2 = synth_integrals(sorb)
        onv = G(d[key + "_onv"])
        ranks = d[key + "_ranks"]
        comb, hm = cx.get_comb_hij_fused(onv, G(h1), G(h2), sorb, noA + noB, noA, noB)
        c, h = comb.cpu().numpy(), hm.cpu().numpy()
        assert sha(c) == str(d[key + "_comb_sha"]), key
        assert sha(h) == str(d[key + "_hmat_sha"]), key
        assert np.array_equal(c[:, ranks], d[key + "_comb"]) and np.array_equal(h[:, ranks], d[key + "_hmat"])
        _, hm32 = cx.get_comb_hij_fused(onv, G(h1).float(), G(h2).float(), sorb, noA + noB, noA, noB)
        assert np.array_equal(hm32.cpu().numpy()[:, ranks], d[key + "_hmat_f32"]), key
        assert np.array_equal(cx.onv_to_tensor(onv, sorb).cpu().numpy(), d[key + "_pm1"])
        sub = comb[0, torch.from_numpy(ranks[:48]).cuda()].contiguous()
        assert np.array_equal(cx.get_hij_torch(sub, sub, G(h1), G(h2), sorb, noA + noB).cpu().numpy(), d[key + "_hij2d"])
        # unfused 3-D path on the sampled columns only (keeps the generic kernel's serial paths cheap)
        sel = comb[:, torch.from_numpy(ranks).cuda()].contiguous()
        assert np.array_equal(cx.get_hij_torch(onv, sel, G(h1), G(h2), sorb, noA + noB).cpu().numpy(), d[key + "_hmat"])
    torch.set_default_dtype(torch.float32)


def test_fe2s2_shipped_problem(cx, fe2s2):
    g = golden("fe2s2_hmat.npz")
    f = fe2s2
    x = G(f["ci_space"][:64])
    h1, h2 = G(f["h1e"]), G(f["h2e"])
    comb, hm = cx.get_comb_hij_fused(x, h1, h2, 40, 30, 15, 15)
    c, h = comb.cpu().numpy(), hm.cpu().numpy()
    assert c.shape == (64, 7876, 8) and h.shape == (64, 7876)
    assert np.array_equal(h[:8], g["hmat8"])
    assert [sha(c[i]) for i in range(64)] == g["comb_sha"].tolist()
    assert [sha(h[i]) for i in range(64)] == g["hmat_sha"].tolist()
    _, hm32 = cx.get_comb_hij_fused(x[:8].contiguous(), h1.float(), h2.float(), 40, 30, 15, 15)
    assert np.array_equal(hm32.cpu().numpy(), g["hmat8_f32"])
    # fused == unfused, like cpp_src/test/hij_float32_float64.py
    assert torch.equal(cx.get_hij_torch(x, comb, h1, h2, 40, 30), hm)
    c2, _ = cx.get_comb_tensor(x, 40, 30, 15, 15)
    assert torch.equal(c2, comb)


def test_wavefunction_lut(cx):
    d = golden("wavefunction_lut.npz")
    for sorb in (40, 100, 184):
        k = f"s{sorb}"
        idx, mask = cx.wavefunction_lut(G(d[k + "_keys"]), G(d[k + "_query"]), sorb)
        assert idx.is_cuda and mask.dtype == torch.bool
        assert np.array_equal(idx.cpu().numpy(), d[k + "_idx"]) and np.array_equal(mask.cpu().numpy(), d[k + "_mask"])
        i2, m2 = cx.wavefunction_lut(torch.from_numpy(d[k + "_keys"]), G(d[k + "_query"]), sorb)
        assert i2.device.type == "cpu" and np.array_equal(i2.numpy(), d[k + "_idx"])
        # the hash table returns the same positions
        ht = cx.hash_build(G(d[k + "_keys"]), sorb)
        i3, m3 = cx.hash_lookup(ht, G(d[k + "_query"]))
        assert np.array_equal(i3.cpu().numpy(), d[k + "_idx"]) and np.array_equal(m3.cpu().numpy(), d[k + "_mask"])
        assert ht.memory >= 4 * d[k + "_keys"].shape[0] * 16


def test_random_against_oracle(cx):
    """Seeded random walkers / dense integrals at sizes the oracle finishes in seconds."""
    from oracle import oracle as O

    for (sorb, noA, noB, n) in [(56, 7, 7, 24), (40, 15, 15, 40), (120, 8, 7, 6), (184, 5, 6, 4), (30, 0, 3, 5), (30, 15, 1, 3)]:
        h1, h2 = synth_integrals(sorb, seed=99)
        onv_np = O.pm01_to_onv(rand_occ(n, sorb, noA, noB, seed=sorb), sorb)
        for dt in (np.float64, np.float32):
            co, ho = O.comb_hij_fused(onv_np, h1.astype(dt), h2.astype(dt), sorb, noA + noB, noA, noB)
            comb, hm = cx.get_comb_hij_fused(G(onv_np), G(h1.astype(dt)), G(h2.astype(dt)), sorb, noA + noB, noA, noB)
            assert np.array_equal(comb.cpu().numpy(), co), (sorb, dt)
            assert np.array_equal(hm.cpu().numpy(), ho), (sorb, dt)


def test_cpu_tensors_are_staged_through_the_gpu(cx, fe2s2):
    f = fe2s2
    x = torch.from_numpy(f["ci_space"][:4].copy())
    comb, hm = cx.get_comb_hij_fused(x, torch.from_numpy(f["h1e"]), torch.from_numpy(f["h2e"]), 40, 30, 15, 15)
    assert comb.device.type == "cpu" and hm.device.type == "cpu"
    g = golden("fe2s2_hmat.npz")
    assert np.array_equal(hm.numpy(), g["hmat8"][:4])


def test_edge_cases_and_errors(cx, fe2s2):
    f = fe2s2
    h1, h2 = G(f["h1e"]), G(f["h2e"])
    e = torch.empty((0, 8), dtype=torch.uint8).cuda()
    comb, hm = cx.get_comb_hij_fused(e, h1, h2, 40, 30, 15, 15)
    assert comb.shape == (0, 7876, 8) and hm.shape == (0, 7876)  # cpu_tensor.cpp:230-237
    assert cx.tensor_to_onv(torch.empty((0, 40), dtype=torch.uint8).cuda(), 40).shape == (0, 8)
    assert cx.onv_to_tensor(e, 40).shape == (0, 40)
    x = G(f["ci_space"][:4])
    with pytest.raises(RuntimeError):
        cx.get_comb_hij_fused(x.t(), h1, h2, 40, 30, 15, 15)  # not contiguous
    with pytest.raises(RuntimeError):
        cx.get_comb_hij_fused(x.to(torch.int32), h1, h2, 40, 30, 15, 15)  # not uint8
    with pytest.raises(RuntimeError):
        cx.get_comb_hij_fused(x, h1, h2, 100, 30, 15, 15)  # 8*bra_len != size(-1)
    with pytest.raises(ValueError):
        cx.check_sorb(200, 10)
    with pytest.raises(OverflowError):
        cx.check_sorb(184, 130)
    cx.check_sorb(56, 14)  # accepted here (run-time word count); the reference's L=1 build rejects it


@pytest.mark.parametrize("sorb", [8, 40, 64, 72, 128, 184, 192, 12, 66])
def test_tensor_to_onv_aligned_and_unaligned_rows(sorb):
    """tensor_to_onv (cpu_tensor.cpp:8-44): 1 = occupied, ANY other byte = empty; the 8-bytes-per-load kernel
    (sorb % 8 == 0) and the byte kernel must both equal a numpy packing, also from an unaligned view."""
    from pynqs_amd import C_extension as cx

    g = np.random.default_rng(sorb)
    occ = g.choice(np.array([0, 1, 1, 2, 255], dtype=np.uint8), size=(257, sorb))
    L = (sorb - 1) // 64 + 1
    bits = np.zeros((257, 64 * L), dtype=np.uint8)
    bits[:, :sorb] = occ == 1
    want = np.packbits(bits.reshape(257, 8 * L, 8), axis=-1, bitorder="little").reshape(257, 8 * L)
    got = cx.tensor_to_onv(torch.from_numpy(occ).cuda(), sorb)
    assert np.array_equal(got.cpu().numpy(), want)
    flat = torch.zeros(257 * sorb + 3, dtype=torch.uint8, device="cuda")
    flat[3:] = torch.from_numpy(occ).cuda().reshape(-1)
    got2 = cx.tensor_to_onv(flat[3:].view(257, sorb), sorb)  # storage offset 3: not 8-byte aligned
    assert np.array_equal(got2.cpu().numpy(), want)


def test_get_hij_on_the_list_just_enumerated_takes_the_plan_kernel(cx, fe2s2):
    """The reference's REDUCE / SAMPLE_SPACE code calls get_comb_tensor(x) and then get_hij_torch(x, comb_x) (eloc.py:243-252,370-378):
    that pair of tensor objects is recognised and answered by the fused plan kernel (Hmat only); a copy of the list, a list written to
    afterwards, or another bra go through the generic pair kernel.  All bit-identical, f64 and f32."""
    import torch

    dev = torch.device("cuda")
    x = torch.from_numpy(np.ascontiguousarray(fe2s2["ci_space"][:64])).to(dev)
    for dt in (torch.float64, torch.float32):
        h1, h2 = torch.from_numpy(fe2s2["h1e"]).to(dev).to(dt), torch.from_numpy(fe2s2["h2e"]).to(dev).to(dt)
        comb, _ = cx.get_comb_tensor(x, 40, 30, 15, 15)
        assert cx._is_last_comb(x, comb, 40, 30) == (15, 15)
        fast = cx.get_hij_torch(x, comb, h1, h2, 40, 30)
        copy = comb.clone()
        assert cx._is_last_comb(x, copy, 40, 30) is None
        slow = cx.get_hij_torch(x, copy, h1, h2, 40, 30)
        assert torch.equal(fast, slow)
        _, hm = cx.get_comb_hij_fused(x, h1, h2, 40, 30, 15, 15)
        assert torch.equal(fast, hm)
    comb, _ = cx.get_comb_tensor(x, 40, 30, 15, 15)
    comb[3, 5, 0] ^= 3  # written to: no longer "the list of x"
    assert cx._is_last_comb(x, comb, 40, 30) is None
    h1, h2 = torch.from_numpy(fe2s2["h1e"]).to(dev), torch.from_numpy(fe2s2["h2e"]).to(dev)
    got = cx.get_hij_torch(x, comb, h1, h2, 40, 30)
    assert torch.equal(got, cx.get_hij_torch(x, comb.clone(), h1, h2, 40, 30)) and not torch.equal(got, hm.double())
    y = x.clone()
    comb, _ = cx.get_comb_tensor(x, 40, 30, 15, 15)
    assert cx._is_last_comb(y, comb, 40, 30) is None

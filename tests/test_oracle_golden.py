"""The oracle (oracle/liboracle.so, a CPU restatement) against the golden vectors captured from the
compiled reference (tests/golden/make_golden.py).  Bit-exact everywhere: comb, Hmat f64 AND f32."""
import hashlib

import numpy as np
import pytest

from conftest import golden, synth_integrals
from oracle import oracle as O


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_docstring_examples():
    d = golden("docstring_examples.npz")
    bra = np.array([[0b1100, 0, 0, 0, 0, 0, 0, 0]], dtype=np.uint8)
    comb, pm = O.comb(bra, 4, 1, 1, flag_bit=True)
    assert np.array_equal(comb, d["comb_onv"]) and np.array_equal(pm, d["comb_states"])
    assert comb[0, :, 0].tolist() == [12, 9, 6, 3]  # libs/C_extension.pyi:85-88
    assert np.array_equal(O.pm01_to_onv(np.array([1, 1, 1, 1, 0, 0, 0, 0], dtype=np.uint8), 8), d["t2o"])
    assert np.array_equal(O.onv_to_pm1(np.array([[15, 0, 0, 0, 0, 0, 0, 0]], dtype=np.uint8), 8), d["o2t"])


def test_c1_exhaustive():
    d = golden("c1_sorb8_all36.npz")
    sorb, noA, noB = int(d["sorb"]), int(d["noA"]), int(d["noB"])
    onv = O.pm01_to_onv(d["occ"], sorb)
    assert np.array_equal(onv, d["onv"])
    comb, hm = O.comb_hij_fused(onv, d["h1e"], d["h2e"], sorb, noA + noB, noA, noB)
    assert np.array_equal(comb, d["comb"]) and np.array_equal(hm, d["hmat"])
    _, hm32 = O.comb_hij_fused(onv, d["h1e"].astype(np.float32), d["h2e"].astype(np.float32), sorb, noA + noB, noA, noB)
    assert np.array_equal(hm32, d["hmat_f32"])
    c2, pm = O.comb(onv, sorb, noA, noB, True)
    assert np.array_equal(c2, d["comb"]) and np.array_equal(pm, d["comb_pm1"])
    assert np.array_equal(O.hij(onv, comb, d["h1e"], d["h2e"], sorb, noA + noB), d["hmat"])
    h2d = O.hij(onv, onv, d["h1e"], d["h2e"], sorb, noA + noB)
    assert np.array_equal(h2d, d["hij2d"])
    assert np.array_equal(h2d, h2d.T)  # real symmetric H
    assert np.array_equal(O.hij(onv, onv, d["h1e"].astype(np.float32), d["h2e"].astype(np.float32), sorb, noA + noB),
                          d["hij2d_f32"])
    assert np.array_equal(O.onv_to_pm1(onv, sorb), d["pm1"])
    assert np.array_equal(O.onv_to_pm1(onv, sorb, np.float32), d["pm1_f32"])


def _cases(fname):
    d = golden(fname)
    keys = sorted({k.rsplit("_", 1)[0] for k in d.files if k.endswith("_onv")})
    return d, keys


def test_asymmetric_small():
    d, keys = _cases("asym_small.npz")
    assert len(keys) == 7
    for key in keys:
        sorb, noA, noB = (int(t[1:]) for t in key.split("_"))
        h1, h2 = synth_integrals(sorb)
        comb, hm = O.comb_hij_fused(d[key + "_onv"], h1, h2, sorb, noA + noB, noA, noB)
        assert np.array_equal(comb, d[key + "_comb"]), key
        assert np.array_equal(hm, d[key + "_hmat"]), key
        _, hm32 = O.comb_hij_fused(d[key + "_onv"], h1.astype(np.float32), h2.astype(np.float32), sorb, noA + noB, noA, noB)
        assert np.array_equal(hm32, d[key + "_hmat_f32"]), key
        # every ket conserves N_alpha, N_beta and is distinct
        w = comb.view(np.uint64)[..., 0]
        assert all(len(set(row.tolist())) == row.size for row in w)


def test_word_boundaries():
    d, keys = _cases("word_boundary.npz")
    assert len(keys) == 8
    for key in keys:
        sorb, noA, noB = (int(t[1:]) for t in key.split("_"))
        h1, h2 = synth_integrals(sorb)
        onv = d[key + "_onv"]
        ranks = d[key + "_ranks"]
        comb, hm = O.comb_hij_fused(onv, h1, h2, sorb, noA + noB, noA, noB)
        assert sha(comb) == str(d[key + "_comb_sha"]) and sha(hm) == str(d[key + "_hmat_sha"]), key
        assert np.array_equal(comb[:, ranks], d[key + "_comb"]) and np.array_equal(hm[:, ranks], d[key + "_hmat"])
        _, hm32 = O.comb_hij_fused(onv, h1.astype(np.float32), h2.astype(np.float32), sorb, noA + noB, noA, noB)
        assert np.array_equal(hm32[:, ranks], d[key + "_hmat_f32"])
        assert np.array_equal(O.onv_to_pm1(onv, sorb), d[key + "_pm1"])
        sub = np.ascontiguousarray(comb[0, ranks[:48]])
        assert np.array_equal(O.hij(sub, sub, h1, h2, sorb, noA + noB), d[key + "_hij2d"])


def test_fe2s2_shipped_problem(fe2s2):
    g = golden("fe2s2_hmat.npz")
    f = fe2s2
    assert (f["sorb"], f["nele"], f["noA"], f["noB"]) == (40, 30, 15, 15)
    x = np.ascontiguousarray(f["ci_space"][:64])
    # SURVEY.md App. B known answers
    assert x.view(np.uint64)[:4, 0].tolist() == [1073741823, 2684354559, 9126805503, 3087007743]
    comb, hm = O.comb_hij_fused(x, f["h1e"], f["h2e"], 40, 30, 15, 15)
    assert comb.shape == (64, 7876, 8)
    assert np.array_equal(hm[:8], g["hmat8"]) and sha(comb[:8]) == str(g["comb8_sha"])
    assert [sha(comb[i]) for i in range(64)] == g["comb_sha"].tolist()
    assert [sha(hm[i]) for i in range(64)] == g["hmat_sha"].tolist()
    np.testing.assert_allclose(hm[0, :5], [-108.489589907013851, 0.010074773570021, 0.007142855790030,
                                           -0.023568691540096, 0.010835253537886], rtol=0, atol=1e-14)
    _, hm32 = O.comb_hij_fused(x[:8], f["h1e"].astype(np.float32), f["h2e"].astype(np.float32), 40, 30, 15, 15)
    assert np.array_equal(hm32, g["hmat8_f32"])
    assert sha(comb[:4])[:16] == "2b51e2d61b54f004" and sha(hm[:4])[:16] == "06c80c3260d7dbbb"


def test_wavefunction_lut():
    d = golden("wavefunction_lut.npz")
    for sorb in (40, 100, 184):
        k = f"s{sorb}"
        idx, mask = O.wavefunction_lut(d[k + "_keys"], d[k + "_query"], sorb)
        assert np.array_equal(idx, d[k + "_idx"]) and np.array_equal(mask, d[k + "_mask"])
        assert mask.sum() >= d[k + "_keys"].shape[0] // 3 and (~mask).any()
        assert np.array_equal(O.sort_keys(d[k + "_keys"], sorb), np.arange(d[k + "_keys"].shape[0]))


def test_integral_layout_roundtrip():
    sorb = 6
    h1, h2 = synth_integrals(sorb)
    a, b = O.decompress_h1e_h2e(h1, h2, sorb)
    # antisymmetry of the decompressed tensor and exact round trip
    assert np.array_equal(b, -b.transpose(1, 0, 2, 3)) and np.array_equal(b, -b.transpose(0, 1, 3, 2))
    assert np.array_equal(b, b.transpose(2, 3, 0, 1))
    c, e = O.compress_h1e_h2e(a, b, sorb)
    assert np.array_equal(c, h1) and np.array_equal(e, h2)
    with pytest.raises(ValueError):
        O.decompress_h1e_h2e(h1[:-1], h2, sorb)


def test_eloc_end_to_end_against_reference_python(fe2s2):
    """vmc/energy/eloc.py (_simple / _only_sample_space) outputs captured from the reference."""
    d = golden("eloc_e2e_fe2s2.npz")
    f = fe2s2
    x = d["x"][:8]
    e, p0 = O.eloc_simple_rbm(x, f["h1e"], f["h2e"], 40, 30, 15, 15, d["W"], d["hb"], d["vb"])
    np.testing.assert_allclose(p0, d["psi_simple"][:8], rtol=1e-12)
    np.testing.assert_allclose(e, d["eloc_simple"][:8], rtol=0, atol=1e-8)  # north-star tolerance: 1e-8 Ha
    np.testing.assert_allclose(O.rbm_real_psi(d["psi_lut_keys"][:64], 40, d["W"], d["hb"], d["vb"]),
                               d["psi_lut"][:64], rtol=1e-12)
    order = O.sort_keys(d["psi_lut_keys"], 40)
    keys = np.ascontiguousarray(d["psi_lut_keys"][order])
    e, p0 = O.eloc_sample_space(d["x"], f["h1e"], f["h2e"], 40, 30, 15, 15, keys, d["psi_lut"][order])
    np.testing.assert_allclose(e, d["eloc_sample_space"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(p0, d["psi_sample_space"], rtol=1e-14)
    e, p0 = O.eloc_sample_space(d["x"], f["h1e"], f["h2e"], 40, 30, 15, 15, keys, d["psi_lut_c"][order])
    np.testing.assert_allclose(e, d["eloc_sample_space_c"], rtol=0, atol=1e-8)

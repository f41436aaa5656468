"""CPU-only checks: the C ABI library loads and exports every symbol include/pynqs_amd.h declares, host-side
entry points (no GPU needed) behave like the reference, and the product refuses to compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, synth_integrals
from oracle import oracle as O


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pynqs_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pynqs_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from pynqs_amd import _native

    syms = _declared_symbols()
    assert len(syms) >= 16
    lib = ctypes.CDLL(_native.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pynqs_amd.h but not exported"
    assert sorted(_native.SIGNATURES) == syms, "pynqs_amd/_native.py must bind exactly the declared ABI"
    assert _native.lib().pynqs_abi_version() == 1


def test_host_entry_points():
    from pynqs_amd import C_extension as cx
    from pynqs_amd import _native as N

    # get_Num_SinglesDoubles: SURVEY.md 8 table
    for (sorb, a, b, want) in [(8, 2, 2, 26), (56, 7, 7, 30723), (40, 15, 15, 7875), (120, 30, 30, 1190250), (184, 46, 46, 6624138)]:
        assert cx.get_Num_SinglesDoubles(sorb, a, b) == want == O.num_sd(sorb, a, b)
    cx.check_sorb(40, 30); cx.check_sorb(192, 120); cx.check_sorb(56, 14)
    with pytest.raises(ValueError):
        cx.check_sorb(193, 4)
    with pytest.raises(OverflowError):
        cx.check_sorb(130, 121)
    with pytest.raises(OverflowError):
        cx.check_sorb(190, 60)  # 130 virtual orbitals
    assert (cx.MAX_SORB, cx.MAX_SORB_LEN, cx.MAX_NELE) == (192, 3, 120)
    assert N.lib().pynqs_plan_bytes(40, N.PYNQS_F64) == 8 * (20**4 + 2 * 190**2 + 2 * 400 * 40 + 800 + 1600 + 40)
    assert N.lib().pynqs_plan_bytes(41, N.PYNQS_F64) == -1
    # round 3's host-only size functions (include/pynqs_amd.h)
    lib = N.lib()
    assert lib.pynqs_keys_index_bytes(65536, 120) == 65536 * 5 * 12 and lib.pynqs_keys_index_bytes(0, 40) == 0
    assert lib.pynqs_keys_index_bytes(10, 41) == -1 and lib.pynqs_keys_index_bytes(1 << 27, 40) == -1
    assert lib.pynqs_keys_index_workspace(0, 40) == 0 and lib.pynqs_keys_index_workspace(1000, 40) >= 1000 * 5 * 12
    # factor table of the children forward: (2 sorb + 1) rows of (H + 2 made odd) entries, + the parents, + the flag
    assert lib.pynqs_rbm_children_table_bytes(8192, 40, 40, N.RBM_COMPLEX) == 8192 * 42 * 16 + 81 * 43 * 16 + 8
    assert lib.pynqs_rbm_children_table_bytes(10, 40, 80, N.RBM_REAL) == 10 * 82 * 8 + 81 * 83 * 8 + 8
    assert lib.pynqs_rbm_forward_children_supported(40, 40, N.RBM_COMPLEX) == 1 and lib.pynqs_rbm_forward_children_supported(40, 80, N.RBM_REAL) == 1
    assert lib.pynqs_rbm_forward_children_supported(120, 240, N.RBM_REAL) == 1 and lib.pynqs_rbm_forward_children_supported(40, 40, 7) == 0  # (beyond the LDS: a wave per row)
    assert lib.pynqs_rbm_grad_workspace(8192, 40, 40, N.RBM_COMPLEX) == 256 * (40 * 41 + 41) * 16
    assert lib.pynqs_rbm_grad_workspace(33, 40, 80, N.RBM_REAL) == 2 * (80 * 41 + 41) * 8 and lib.pynqs_rbm_grad_workspace(8, 40, 40, N.RBM_TANH) == -1
    # the complex-parameter RBM kernel: windowed beyond the LDS, refused only when the per-hidden-unit arrays alone do not fit
    assert lib.pynqs_eloc_crbm_supported(40, 30, 15, 15, 80) == 1 and lib.pynqs_eloc_crbm_supported(120, 60, 30, 30, 240) == 1
    assert lib.pynqs_eloc_crbm_supported(66, 4, 2, 2, 4000) == 0


def test_integral_layout_matches_oracle():
    from pynqs_amd import C_extension as cx

    sorb = 8
    h1, h2 = synth_integrals(sorb)
    a, b = cx.decompress_h1e_h2e(h1, h2, sorb)
    ao, bo = O.decompress_h1e_h2e(h1, h2, sorb)
    assert np.array_equal(a, ao) and np.array_equal(b, bo)
    c, e = cx.compress_h1e_h2e(a, b, sorb)
    assert np.array_equal(c, h1) and np.array_equal(e, h2)
    # non-antisymmetric input: the last writer of a slot wins, as in integral.cpp:45-53
    g = np.random.default_rng(5).standard_normal((sorb,) * 4)
    c1, e1 = cx.compress_h1e_h2e(a, g, sorb)
    c2, e2 = O.compress_h1e_h2e(a, g, sorb)
    assert np.array_equal(e1, e2)
    with pytest.raises(ValueError):
        cx.decompress_h1e_h2e(h1[:-1], h2, sorb)


def test_no_cpu_fallback():
    """Without a HIP device every compute entry point raises instead of silently computing on the host."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pynqs_amd import C_extension as cx

    x = torch.zeros((2, 8), dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="no HIP device"):
        cx.onv_to_tensor(x, 40)
    with pytest.raises(RuntimeError, match="no HIP device"):
        cx.get_comb_tensor(x, 40, 30, 15, 15)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pynqs_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle's", "").replace("the oracle", "").lower() or f == "build.py", f


def test_split_helpers_and_sorting():
    from pynqs_amd import public_function as pf

    assert pf.split_batch_idx(11, 3) == [3, 6, 9, 11]      # utils/public_function.py docstring
    assert pf.split_length_idx(11, 3) == [4, 8, 11]
    d = golden("wavefunction_lut.npz")
    for sorb in (40, 100, 184):
        keys = torch.from_numpy(d[f"s{sorb}_keys"])
        perm = torch.randperm(keys.size(0), generator=torch.Generator().manual_seed(1))
        idx = pf.torch_sort_onv(keys[perm])
        assert torch.equal(keys[perm][idx], keys)
    bra = torch.tensor([[3, 0, 0, 0, 0, 0, 0, 0], [12, 0, 0, 0, 0, 0, 0, 0], [9, 0, 0, 0, 0, 0, 0, 0], [6, 0, 0, 0, 0, 0, 0, 0]],
                       dtype=torch.uint8)
    assert pf.torch_sort_onv(bra).tolist() == [0, 3, 2, 1]
    # spin-flip helpers, uint8 form vs occupation form
    occ = torch.tensor([[1, 1, 0, 1, 1, 0, 1, 1], [1, 0, 0, 1, 1, 1, 0, 0]], dtype=torch.int64)
    byte = torch.tensor([[0b11011011, 0, 0, 0, 0, 0, 0, 0], [0b00111001, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.uint8)
    assert pf.spin_flip_sign(byte, 8).tolist() == pf.spin_flip_sign(occ, 8).tolist() == [1, -1]
    assert pf.spin_flip_onv(byte, 8)[:, 0].tolist() == [0b11100111, 0b00110110]


def test_unique_onv_matches_torch_unique():
    """Word-wise unique of onv rows = torch.unique(dim=0) up to the order of the unique rows (1-3 words, empty)."""
    import torch
    from pynqs_amd.public_function import unique_onv

    g = torch.Generator().manual_seed(3)
    for L in (1, 2, 3):
        x = torch.randint(0, 256, (64, 8 * L), dtype=torch.uint8, generator=g)
        x = x[torch.randint(0, 64, (1000,), generator=g)]  # many duplicates
        u, inv = unique_onv(x)
        ref, _ = torch.unique(x, dim=0, return_inverse=True)
        assert torch.equal(u[inv], x)
        assert u.size(0) == ref.size(0)
        assert torch.equal(torch.unique(u, dim=0), ref)
    u, inv = unique_onv(torch.empty((0, 16), dtype=torch.uint8))
    assert u.shape == (0, 16) and inv.numel() == 0


def test_header_is_valid_c99(tmp_path):
    """include/pynqs_amd.h is the C ABI: it must compile as plain C (no C++-isms), and a C translation unit that
    takes the address of every declared entry point must type-check."""
    import re
    import subprocess

    from pynqs_amd import _native

    hdr = os.path.join(ROOT, "include", "pynqs_amd.h")
    src = tmp_path / "abi.c"
    names = sorted(_native.SIGNATURES)
    src.write_text('#include "pynqs_amd.h"\nvoid *table[] = {\n' + "".join(f"  (void *){n},\n" for n in names) + "};\n")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-Wno-pedantic", "-fsyntax-only", "-I", os.path.dirname(hdr), str(src)])
    declared = set(re.findall(r"\b(pynqs_[a-z0-9_]+)\s*\(", open(hdr).read()))
    assert declared == set(names)


def test_dropin_matches_the_reference_api_surface():
    """Every name of the reference's `libs.C_extension` (libs/C_extension.pyi; captured as data by tests/golden/make_api_fixture.py)
    exists in the drop-in module with the same parameter names, order and defaults; stubs of functions outside the local-energy path
    raise NotImplementedError."""
    import inspect
    import json
    import os

    import pytest

    from pynqs_amd.dropin.libs import C_extension as drop

    api = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c_extension_api.json")))
    out_of_path = {"MCMC_sample", "mps_vbatch", "convert_sites"}
    for name, spec in api["functions"].items():
        assert hasattr(drop, name), name
        fn = getattr(drop, name)
        if name in out_of_path:
            with pytest.raises(NotImplementedError):
                fn(*([None] * len(spec["params"])))
            continue
        sig = inspect.signature(fn)
        params = list(sig.parameters.values())
        got = [q.name for q in params]
        # (the drop-in may add trailing keyword parameters with defaults, never rename or reorder the reference's)
        assert got[:len(spec["params"])] == spec["params"], (name, got, spec["params"])
        for q in params[len(spec["params"]):]:
            assert q.default is not inspect.Parameter.empty, (name, q.name)
        for pn, dv in spec["defaults"].items():
            assert sig.parameters[pn].default == dv, (name, pn)
    for cls, methods in api["classes"].items():
        assert hasattr(drop, cls), cls
        for m in methods:
            assert hasattr(getattr(drop, cls), m), (cls, m)
    for attr in api["attributes"]:
        assert isinstance(getattr(drop, attr), int), attr


def test_ansatz_side_helpers_match_the_reference_extension():
    """permute_sgn (cpu_tensor.cpp:356 -> onstate.cpp:195) and constrain_make_charts (cpu_tensor.cpp:558): plain tensor algebra in the
    drop-in (the reference's autoregressive ansaetze call them), against outputs of the compiled reference extension
    (tests/golden/ansatz_helpers.npz)."""
    import numpy as np
    import torch

    from conftest import golden
    from pynqs_amd.dropin.libs import C_extension as drop

    d = golden("ansatz_helpers.npz")
    for sorb in (8, 40):
        occ = torch.from_numpy(d[f"occ_{sorb}"])
        for perm, ref in zip(d[f"perm_{sorb}"], d[f"sgn_{sorb}"]):
            got = drop.permute_sgn(torch.from_numpy(perm), occ, sorb)
            assert got.dtype == torch.float64 and np.array_equal(got.numpy(), ref)
    assert drop.permute_sgn(torch.arange(8), torch.zeros((0, 8), dtype=torch.int64), 8).shape == (0,)
    got = drop.constrain_make_charts(torch.from_numpy(d["chart_idx"]))
    assert got.dtype == torch.float64 and np.array_equal(got.numpy(), d["charts"])
    assert drop.constrain_make_charts(torch.zeros(0, dtype=torch.int64)).shape == (0, 4)


def test_tensors_without_version_counters_are_not_tracked():
    """Plans and the last S+D list are cached by tensor identity + version counter; tensors created under torch.inference_mode()
    have no counter (reading it raises): they count as untrackable instead of breaking the call."""
    import torch

    from pynqs_amd import C_extension as cx

    with torch.inference_mode():
        t = torch.zeros(4)
    assert cx._ver(t) == -1 and cx._ver(torch.zeros(4)) == 0
    cx._remember_comb(t, t, 8, 4, 2, 2)  # CPU tensors: nothing is remembered, nothing raises
    assert cx._is_last_comb(t, t, 8, 4) is None


def test_spin_projection_follows_the_host_programs_instance():
    """With pynqs_amd.energy imported into a PyNQS process the run calls PyNQS' own SpinProjection.init (utils/public_function.py:1017-1036);
    an uninitialised pynqs_amd instance reads eta from there."""
    import sys
    import types

    import pytest

    from pynqs_amd import public_function as pf

    mine = pf._SpinProjection()
    with pytest.raises(NotImplementedError):
        mine.eta
    host = types.ModuleType("utils.public_function")
    host.SpinProjection = pf._SpinProjection()
    sys.modules["utils.public_function"] = host
    try:
        with pytest.raises(NotImplementedError):
            mine.eta  # the host's is not initialised either
        host.SpinProjection.init(30, 1)
        assert mine.eta == host.SpinProjection.eta == (-1) ** (15 - 1)
        mine.init(30, 0)
        assert mine.eta == -1  # its own initialisation wins
    finally:
        del sys.modules["utils.public_function"]


def check_integral_layout(sorb: int) -> None:
    """compress / decompress of the integrals (cpp_src/tensor/integral.cpp:6-125) against the oracle, with the working memory beyond
    input and output bounded by one [s, s, s] block (the first version built several s^4 temporaries: 13 GB at sorb 120)."""
    import tracemalloc

    from pynqs_amd import C_extension as cx

    h1, h2 = synth_integrals(sorb)
    tracemalloc.start()
    a, b = cx.decompress_h1e_h2e(h1, h2, sorb)
    _, peak = tracemalloc.get_traced_memory()
    tracemalloc.stop()
    assert peak < b.nbytes + 40 * sorb**3 + 2**26, f"decompress peaked at {peak / 2**30:.2f} GiB for a {b.nbytes / 2**30:.2f} GiB result"
    ao, bo = O.decompress_h1e_h2e(h1, h2, sorb)
    assert np.array_equal(a, ao) and np.array_equal(b, bo)
    del ao, bo
    tracemalloc.start()
    c, e = cx.compress_h1e_h2e(a, b, sorb)
    _, peak = tracemalloc.get_traced_memory()
    tracemalloc.stop()
    assert peak < e.nbytes + 40 * sorb**3 + 2**26, f"compress peaked at {peak / 2**30:.2f} GiB"
    assert np.array_equal(c, h1) and np.array_equal(e, h2)  # round trip: every packed element comes back
    co, eo = O.compress_h1e_h2e(a, b, sorb)
    assert np.array_equal(c, co) and np.array_equal(e, eo)


def test_integral_layout_in_blocks():
    """sorb 64 here (134 MB full tensor); sorb 120 (1.7 GB) runs in the GPU tier, tests/test_gpu_misc_r3.py: first-touch of gigabyte
    arrays takes minutes in the development container (the oracle's C loop just the same), seconds on an ordinary host."""
    check_integral_layout(64)


def test_get_nbatch_matches_the_reference_and_knows_the_fused_paths():
    """utils/public_function.py:162-261.  Expected values: the reference's own get_nbatch on the CPU device for these arguments
    (captured in the development container; inputs and outputs only)."""
    from pynqs_amd import public_function as pf

    cpu = torch.device("cpu")
    cases = [(40, 100000, 7875, 32, 0.25, False, torch.double), (40, 1000, 7875, 32, 0.25, False, torch.double),
             (120, 50000, 1190250, 64, 0.5, False, torch.double), (40, 100000, 7875, 32, 0.25, True, torch.double),
             (40, 18496, 7875, 32, 0.25, True, torch.complex128), (120, 1000000, 1190250, 48, 2.0, True, torch.complex128),
             (184, 32768, 6624138, 200, 0.25, True, torch.double)]
    want = [1704, 1000, 15, 100000, 18496, 3221, 32768]
    got = [pf.get_nbatch(s, n, nsd, mm, a, device=cpu, use_sample=us, dtype=dt) for (s, n, nsd, mm, a, us, dt) in cases]
    assert got == want
    # fused paths: nothing of size walkers x n_sd is allocated
    assert pf.get_nbatch(40, 65536, 7875, 32, 0.25, cpu, True, torch.complex128, fused="sample_space") == 65536
    assert pf.get_nbatch(120, 10**7, 1190250, 32, 0.25, cpu, False, fused="simple_rbm") == 1 << 22
    r = pf.get_nbatch(40, 10**6, 7875, 32, 0.25, cpu, False, fused="reduce", eps_sample=1000)
    assert 8192 <= r <= 65536  # ~0.45 MB per walker (every record counted as a distinct x') against a quarter of 32 GiB
    assert pf.get_nbatch(120, 10**6, 1190250, 32, 0.25, cpu, False, fused="reduce") < r
    with pytest.raises(ValueError):
        pf.get_nbatch(40, 10, 7875, fused="nope")


def test_reduce_front_list_capacity_is_host_arithmetic():
    """pynqs_reduce_onepass_list_capacity: the kept doubles per segment up to which the one-launch front end runs in one of its LIST forms.
    Short rows: 1024 slots minus the fixed ones; without draws the flushing form extends that to a tenth of a segment's columns, and to any
    capacity without a de-duplication table.  Rows of more than 65536 columns: any capacity (flushing form; with draws it keeps its tile sums
    in io->tile_scratch, which also makes sorb 184 with whole rows possible: pynqs_reduce_onepass_geometry's out[3])."""
    from pynqs_amd import reduce_front as RF

    any_cap = (1 << 30) - 1
    assert RF.list_capacity(8192, 40, 30, 15, 15, 1000) == 1024 - 168 and RF.list_capacity(4096, 56, 14, 7, 7, 1000) == 30976 // 10  # (beyond the cached LIST form: flushing, no cache)
    assert RF.list_capacity(8192, 40, 30, 15, 15, 100) == 1024 - 168 and RF.list_capacity(4096, 56, 14, 7, 7, 200) == 30976 // 10
    assert RF.list_capacity(8192, 40, 30, 15, 15, 100, without_table=True) == any_cap
    assert RF.list_capacity(8192, 40, 30, 15, 15, 0) == 1024 - 168          # (7936 columns: a tenth is less than the list)
    assert RF.list_capacity(4096, 56, 14, 7, 7, 0) == 30976 // 10            # (sorb 56: 30724 columns in one segment of 30976)
    for args in ((8192, 40, 30, 15, 15, 0), (4096, 56, 14, 7, 7, 0)):
        assert RF.list_capacity(*args, without_table=True) == any_cap
    for args in ((2048, 80, 40, 20, 20, 0), (4096, 120, 60, 30, 30, 0), (256, 120, 60, 30, 30, 0), (4096, 184, 92, 46, 46, 0),
                 (2048, 80, 40, 20, 20, 1000), (4096, 120, 60, 30, 30, 1000), (4096, 184, 92, 46, 46, 1000)):
        assert RF.list_capacity(*args) == any_cap and RF.supported(*args)
    assert int(N_lib().pynqs_reduce_onepass_tile_scratch_bytes(10, 120, 60, 30, 30, 0)) == 0
    assert int(N_lib().pynqs_reduce_onepass_tile_scratch_bytes(10, 120, 60, 30, 30, 1000)) % 160 == 0 > -1


def N_lib():
    from pynqs_amd import _native as N

    return N.lib()

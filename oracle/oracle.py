"""numpy/ctypes front-end of the test oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- the checker for the HIP product.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; nothing under
pynqs_amd/ does.  Parity status: PINNED (see pynqs_oracle.h).

ONVs are numpy uint8 arrays [n, 8*len] (the reference's tensor layout, cpp_src/tensor/
cpu_tensor.cpp:8-44) and are reinterpreted as little-endian uint64 words.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64, _i32, _vp = C.c_int64, C.c_int, C.c_void_p


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("pynqs_oracle.c", "pynqs_oracle_tmpl.inc", "pynqs_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_num_sd.restype = _i64
    return _LIB


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(_vp)


def bra_len(sorb: int) -> int:
    return (sorb - 1) // 64 + 1


def _words(onv: np.ndarray, sorb: int) -> np.ndarray:
    onv = np.ascontiguousarray(onv, dtype=np.uint8)
    assert onv.shape[-1] == 8 * bra_len(sorb), (onv.shape, sorb)
    return onv.view(np.uint64)


def num_sd(sorb: int, noA: int, noB: int) -> int:
    return int(lib().orc_num_sd(sorb, noA, noB))


def merged(bra: np.ndarray, sorb: int) -> np.ndarray:
    w = _words(bra, sorb).reshape(-1, bra_len(sorb))
    out = np.empty((w.shape[0], sorb), dtype=np.int32)
    for i in range(w.shape[0]):
        lib().orc_merged(_p(w[i]), bra_len(sorb), sorb, _p(out[i]))
    return out


def unpack_table(sorb: int, noA: int, noB: int) -> np.ndarray:
    n = num_sd(sorb, noA, noB)
    out = np.empty((n, 5), dtype=np.int32)
    for r in range(n):
        lib().orc_unpack_sd(sorb, noA, noB, r, _p(out[r]))
    return out


def comb(bra: np.ndarray, sorb: int, noA: int, noB: int, flag_bit: bool = False, nthreads: int = 0):
    w = _words(bra, sorb).reshape(-1, bra_len(sorb))
    n, L = w.shape
    nc = num_sd(sorb, noA, noB) + 1
    out = np.empty((n, nc, L), dtype=np.uint64)
    pm = np.empty((n, nc, sorb), dtype=np.float64) if flag_bit else None
    rc = lib().orc_comb(_p(w), _i64(n), sorb, noA, noB, _p(out), _p(pm), nthreads)
    assert rc == 0
    return out.view(np.uint8).reshape(n, nc, 8 * L), pm


def comb_hij_fused(bra, h1e, h2e, sorb, nele, noA, noB, nthreads: int = 0):
    w = _words(bra, sorb).reshape(-1, bra_len(sorb))
    n, L = w.shape
    nc = num_sd(sorb, noA, noB) + 1
    dt = np.asarray(h1e).dtype
    assert dt in (np.float32, np.float64) and np.asarray(h2e).dtype == dt
    h1e = np.ascontiguousarray(h1e); h2e = np.ascontiguousarray(h2e)
    out = np.empty((n, nc, L), dtype=np.uint64)
    hm = np.empty((n, nc), dtype=dt)
    f = lib().orc_comb_hij_fused_f64 if dt == np.float64 else lib().orc_comb_hij_fused_f32
    rc = f(_p(w), _i64(n), sorb, nele, noA, noB, _p(h1e), _p(h2e), _p(out), _p(hm), nthreads)
    assert rc == 0
    return out.view(np.uint8).reshape(n, nc, 8 * L), hm


def hij(bra, ket, h1e, h2e, sorb, nele, nthreads: int = 0):
    L = bra_len(sorb)
    b = _words(bra, sorb).reshape(-1, L)
    ket = np.ascontiguousarray(ket, dtype=np.uint8)
    is3d = ket.ndim == 3
    k = ket.view(np.uint64)
    n = b.shape[0]
    m = ket.shape[1] if is3d else ket.shape[0]
    dt = np.asarray(h1e).dtype
    h1e = np.ascontiguousarray(h1e); h2e = np.ascontiguousarray(h2e)
    hm = np.empty((n, m), dtype=dt)
    f = lib().orc_hij_f64 if dt == np.float64 else lib().orc_hij_f32
    rc = f(_p(b), _i64(n), _p(k), _i64(m), int(is3d), _p(h1e), _p(h2e), sorb, nele, _p(hm), nthreads)
    assert rc == 0
    return hm


def onv_to_pm1(bra, sorb, dtype=np.float64):
    w = _words(bra, sorb).reshape(-1, bra_len(sorb))
    out = np.empty((w.shape[0], sorb), dtype=dtype)
    f = lib().orc_onv_to_pm1_f64 if np.dtype(dtype) == np.float64 else lib().orc_onv_to_pm1_f32
    f(_p(w), _i64(w.shape[0]), sorb, _p(out))
    return out


def pm01_to_onv(occ, sorb):
    occ = np.ascontiguousarray(occ, dtype=np.uint8).reshape(-1, sorb)
    out = np.empty((occ.shape[0], bra_len(sorb)), dtype=np.uint64)
    lib().orc_pm01_to_onv(_p(occ), _i64(occ.shape[0]), sorb, _p(out))
    return out.view(np.uint8).reshape(occ.shape[0], 8 * bra_len(sorb))


def compress_h1e_h2e(h1e2d, h2e4d, sorb):
    h1 = np.ascontiguousarray(h1e2d, dtype=np.float64); h2 = np.ascontiguousarray(h2e4d, dtype=np.float64)
    pair = sorb * (sorb - 1) // 2
    o1 = np.empty(sorb * sorb); o2 = np.empty(pair * (pair + 1) // 2)
    lib().orc_compress_h1e_h2e(_p(h1), _p(h2), sorb, _p(o1), _p(o2))
    return o1, o2


def decompress_h1e_h2e(h1e, h2e, sorb):
    h1 = np.ascontiguousarray(h1e, dtype=np.float64); h2 = np.ascontiguousarray(h2e, dtype=np.float64)
    pair = sorb * (sorb - 1) // 2
    if h1.size != sorb * sorb or h2.size != pair * (pair + 1) // 2:
        raise ValueError("h1e/h2e array size is incorrect")
    o1 = np.empty((sorb, sorb)); o2 = np.empty((sorb,) * 4)
    lib().orc_decompress_h1e_h2e(_p(h1), _p(h2), sorb, _p(o1), _p(o2))
    return o1, o2


def wavefunction_lut(keys, onv, sorb):
    L = bra_len(sorb)
    k = _words(keys, sorb).reshape(-1, L); q = _words(onv, sorb).reshape(-1, L)
    idx = np.empty(q.shape[0], dtype=np.int64); mask = np.empty(q.shape[0], dtype=np.uint8)
    lib().orc_wavefunction_lut(_p(k), _i64(k.shape[0]), _p(q), _i64(q.shape[0]), L, _p(idx), _p(mask))
    return idx, mask.astype(bool)


def sort_keys(keys: np.ndarray, sorb: int) -> np.ndarray:
    """Order that utils/public_function.py:651 (torch_sort_onv) produces: ascending as a big integer,
    least-significant byte first.  Returns the argsort indices."""
    w = _words(keys, sorb).reshape(-1, bra_len(sorb))
    return np.lexsort(tuple(w[:, k] for k in range(w.shape[1])))


def rbm_real_psi(onv, sorb, W, hb, vb):
    w = _words(onv, sorb).reshape(-1, bra_len(sorb))
    W = np.ascontiguousarray(W, dtype=np.float64); hb = np.ascontiguousarray(hb, dtype=np.float64)
    vb = np.ascontiguousarray(vb, dtype=np.float64)
    assert W.shape == (hb.size, sorb)
    psi = np.empty(w.shape[0])
    lib().orc_rbm_real_psi(_p(w), _i64(w.shape[0]), sorb, hb.size, _p(W), _p(hb), _p(vb), _p(psi))
    return psi


def eloc_simple_rbm(bra, h1e, h2e, sorb, nele, noA, noB, W, hb, vb, nthreads: int = 0):
    w = _words(bra, sorb).reshape(-1, bra_len(sorb))
    h1e = np.ascontiguousarray(h1e, dtype=np.float64); h2e = np.ascontiguousarray(h2e, dtype=np.float64)
    W = np.ascontiguousarray(W, dtype=np.float64); hb = np.ascontiguousarray(hb, dtype=np.float64)
    vb = np.ascontiguousarray(vb, dtype=np.float64)
    e = np.empty(w.shape[0]); p0 = np.empty(w.shape[0])
    rc = lib().orc_eloc_simple_rbm(_p(w), _i64(w.shape[0]), sorb, nele, noA, noB, _p(h1e), _p(h2e), hb.size,
                                   _p(W), _p(hb), _p(vb), _p(e), _p(p0), nthreads)
    assert rc == 0
    return e, p0


def eloc_sample_space(bra, h1e, h2e, sorb, nele, noA, noB, keys_sorted, wf, nthreads: int = 0):
    L = bra_len(sorb)
    w = _words(bra, sorb).reshape(-1, L); k = _words(keys_sorted, sorb).reshape(-1, L)
    h1e = np.ascontiguousarray(h1e, dtype=np.float64); h2e = np.ascontiguousarray(h2e, dtype=np.float64)
    wf = np.ascontiguousarray(wf)
    cplx = np.iscomplexobj(wf)
    wf = wf.astype(np.complex128 if cplx else np.float64)
    e = np.empty(w.shape[0], dtype=wf.dtype); p0 = np.empty(w.shape[0], dtype=wf.dtype)
    rc = lib().orc_eloc_sample_space(_p(w), _i64(w.shape[0]), sorb, nele, noA, noB, _p(h1e), _p(h2e), _p(k),
                                     _i64(k.shape[0]), _p(wf), int(cplx), _p(e), _p(p0), nthreads)
    assert rc == 0
    return e, p0

"""Drop-in replacement for PyNQS' ``libs.C_extension`` hot-path functions on MI355X.

Same names, positional/keyword arguments, shapes, dtypes and error behaviour as the reference's
pybind11 module (cpp_src/tensor/bind.cpp:317-391, libs/C_extension.pyi), implemented on the C ABI of
libpynqs_amd.so (include/pynqs_amd.h).  torch is used for device memory and streams only.

Device rule.  The reference runs its CPU kernels when every input is a CPU tensor.  This package has
no CPU kernels: CPU inputs are staged to the current HIP device, the HIP kernels run, and results are
returned on the CPU, so callers that hold CPU tensors keep working.  Without a GPU (or without the
built library) every compute function raises -- there is no fallback.

Error rule.  The reference mixes TORCH_CHECK (-> RuntimeError) with C asserts that abort
the process; here every such check raises RuntimeError.  check_sorb raises ValueError / OverflowError
like bind.cpp:290-299.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
from torch import Tensor

from . import _native as N

MAX_SORB_LEN = 3  # run-time word-count dispatch; the reference compiles one build per length
MAX_SORB = 64 * MAX_SORB_LEN
MAX_NELE = 120
USE_PLAN = True  # route get_comb_hij_fused through the cached integral plan (False: direct packed-triangle kernels)

__all__ = [
    "tensor_to_onv", "onv_to_tensor", "get_comb_tensor", "get_hij_torch", "get_comb_hij_fused",
    "wavefunction_lut", "hash_build", "hash_lookup", "HashTable", "RBMTable", "eloc_rbm", "merge_rank_sample", "spin_flip_rand", "check_sorb", "compress_h1e_h2e", "decompress_h1e_h2e", "get_Num_SinglesDoubles",
    "MAX_SORB", "MAX_SORB_LEN", "MAX_NELE",
]


# ---- plumbing ----------------------------------------------------------------------------------------
def _bra_len(sorb: int) -> int:
    return (sorb - 1) // 64 + 1


def _gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("pynqs_amd: no HIP device available (this package has no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _stage(*ts: Tensor):
    """Returns (device, tensors on that device, all_cpu)."""
    all_cpu = all(t.device.type == "cpu" for t in ts)
    dev = next((t.device for t in ts if t.device.type == "cuda"), None) or _gpu()
    return dev, [t if t.device == dev else t.to(dev) for t in ts], all_cpu


def _ver(t: Tensor) -> int:
    """version counter of a tensor (bumped by in-place writes); tensors created under torch.inference_mode() have none: -1, which
    callers treat as "cannot be tracked" (no reuse of cached plans / lists for such tensors)"""
    try:
        return t._version
    except RuntimeError:
        return -1


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


def _contig(t: Tensor, name: str) -> None:
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")  # CHECK_CONTIGUOUS, common/default.h:17-18


def _check_onv(t: Tensor, name: str, sorb: int, dims) -> None:
    _contig(t, name)
    if t.dtype != torch.uint8:
        raise RuntimeError(f"{name} must be torch.uint8")  # assert in bind.cpp:13,29,51,74
    if t.dim() not in dims:
        raise RuntimeError(f"{name} must have dim in {dims}, got {t.dim()}")
    if t.size(-1) != 8 * _bra_len(sorb):
        raise RuntimeError(f"{name}: last dim {t.size(-1)} != 8 * bra_len = {8 * _bra_len(sorb)} for sorb = {sorb}")


def _fdtype(h1e: Tensor, h2e: Tensor) -> int:
    if h1e.dtype != h2e.dtype or h1e.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("h1e/h2e must both be float32 or both float64")  # AT_DISPATCH_FLOATING_TYPES
    return N.PYNQS_F64 if h1e.dtype == torch.float64 else N.PYNQS_F32


def _check_integrals(h1e: Tensor, h2e: Tensor, sorb: int) -> None:
    _contig(h1e, "h1e_tensor"); _contig(h2e, "h2e_tensor")
    pair = sorb * (sorb - 1) // 2
    if h1e.numel() != sorb * sorb or h2e.numel() != pair * (pair + 1) // 2:
        # the reference reads out of bounds here; an explicit error is the safe equivalent
        raise RuntimeError(f"h1e/h2e sizes {h1e.numel()}/{h2e.numel()} do not match sorb = {sorb}")


class IntegralPlan:
    """Device-resident spin-blocked re-layout of (h1e, h2e) (include/pynqs_amd.h, 'integral plan').
    Built once per pair of integral tensors; values are copies of h1e/h2e elements, so every kernel that
    reads the plan returns bit-identical numbers to the direct-layout kernels."""

    def __init__(self, h1e: Tensor, h2e: Tensor, sorb: int, device: "torch.device | None" = None):
        code = _fdtype(h1e, h2e)
        _check_integrals(h1e, h2e, sorb)
        nbytes = N.lib().pynqs_plan_bytes(sorb, code)
        if nbytes < 0:
            raise RuntimeError(f"integral plan needs an even sorb in [2, {MAX_SORB}], got {sorb}")
        dev, (a, b), _ = _stage(h1e, h2e)
        if device is not None and device.type == "cuda" and all(t.device.type == "cpu" for t in (h1e, h2e)):
            # host-resident integrals go to the device of the walkers, not to the current device
            dev, a, b = device, h1e.to(device), h2e.to(device)
        self.sorb, self.code, self.dtype, self.device = sorb, code, h1e.dtype, dev
        self.buf = torch.empty(nbytes // h1e.element_size(), dtype=h1e.dtype, device=dev)
        N.check(N.lib().pynqs_plan_build(a.data_ptr(), b.data_ptr(), sorb, code, self.buf.data_ptr(), _stream(dev)), "plan_build")
        # a, b may be staging copies: the build kernel must finish before they are released
        if a is not h1e or b is not h2e:
            torch.cuda.current_stream(dev).synchronize()

    def data_ptr(self) -> int:
        return self.buf.data_ptr()


_PLANS: "list[tuple]" = []  # (weakref(h1e), weakref(h2e), versions, sorb, plan, content fingerprint or None), most recent first
_MAX_PLANS = 4
_PLAN_BUILDS: "dict[tuple, int]" = {}  # (sorb, numel of h2e, device) -> number of plans built; a climbing count = a caller that re-creates the integrals
_PLAN_REBUILD_WARN = 16


def _fingerprint(h1e: Tensor, h2e: Tensor):
    """Content key of a pair of integral tensors: shapes, dtypes, devices and two 64-bit sums over the elements' bit patterns (the plain sum
    and the sum after a multiply-xorshift mix, both modulo 2^64) per tensor.  The plan is a pure function of the integrals' content, so a
    cached plan may serve any tensors with the same key: a caller that re-creates equal integrals on every call (a fresh `.to(device)`, a
    reloaded file) pays two reductions and one read-back instead of a rebuild (ms, up to 1.2 GB written at sorb 184).  None when it cannot
    be taken (stream capture: the read-back synchronises)."""
    if any(t.is_cuda for t in (h1e, h2e)) and torch.cuda.is_current_stream_capturing():
        return None
    out = []
    with torch.no_grad():
        for t in (h1e, h2e):
            c = t.detach().contiguous()
            v = c.view(torch.int64) if c.element_size() == 8 else c.view(torch.int32).to(torch.int64)
            m = (v * -7046029254386353131) ^ (v >> 29)   # (0x9E3779B97F4A7C15 as a signed 64-bit factor; wraps)
            out.append((tuple(t.shape), str(t.dtype), str(t.device), int(v.sum().item()), int(m.sum().item())))
    return tuple(out)


def plan_for(h1e: Tensor, h2e: Tensor, sorb: int, device: "torch.device | None" = None) -> "IntegralPlan | None":
    """Cached IntegralPlan for these tensors (None when sorb is odd -> direct kernels).  The cache is keyed on the tensor objects
    (identity, or any alias of a cached object that is still alive; version counter, storage address) and, when those miss, on the
    integrals' CONTENT (_fingerprint: two reductions and one read-back): a fresh copy of equal integrals re-uses the plan instead of
    rebuilding it.  `device`: where host-resident integrals are staged (the walkers' device)."""
    import weakref

    if sorb % 2 or sorb < 2:
        return None
    ver = (_ver(h1e), _ver(h2e), h1e.data_ptr(), h2e.data_ptr())
    trackable = ver[0] >= 0 and ver[1] >= 0  # (inference-mode tensors have no version counter: in-place edits would go unnoticed)
    def same(ref, t):
        # the cached tensor object itself, or -- while that object is alive, so that its storage cannot have been freed and handed out
        # again -- another tensor object over the same memory (a view, .detach(), a re-wrapped parameter): same address, shape, strides, dtype
        o = ref()
        return o is t or (o is not None and o.data_ptr() == t.data_ptr() and o.shape == t.shape and o.stride() == t.stride() and o.dtype == t.dtype)

    dev_ok = lambda pl: device is None or device.type != "cuda" or pl.device == device or h1e.device.type == "cuda"  # noqa: E731
    for i, (r1, r2, v, s, pl, fp) in enumerate(_PLANS if trackable else ()):
        if same(r1, h1e) and same(r2, h2e) and v == ver and s == sorb and dev_ok(pl):
            if i:
                _PLANS.insert(0, _PLANS.pop(i))
            return pl
    # other tensor objects: the same CONTENT as a cached plan's integrals?  (entries of this system only: nothing is computed otherwise)
    fp = None
    if any(s == sorb and f is not None and f[1][0] == tuple(h2e.shape) for _, _, _, s, _, f in _PLANS):
        fp = _fingerprint(h1e, h2e)
        for i, (r1, r2, v, s, pl, f) in enumerate(_PLANS):
            if fp is not None and f == fp and s == sorb and dev_ok(pl):
                _PLANS.pop(i)
                if trackable:   # (from now on these tensor objects hit by identity)
                    _PLANS.insert(0, (weakref.ref(h1e), weakref.ref(h2e), ver, sorb, pl, fp))
                else:
                    _PLANS.insert(0, (r1, r2, v, s, pl, f))
                return pl
    pl = IntegralPlan(h1e, h2e, sorb, device)
    if trackable:
        if fp is None:
            fp = _fingerprint(h1e, h2e)
        _PLANS.insert(0, (weakref.ref(h1e), weakref.ref(h2e), ver, sorb, pl, fp))
        del _PLANS[_MAX_PLANS:]
    key = (sorb, h2e.numel(), str(pl.device))
    _PLAN_BUILDS[key] = _PLAN_BUILDS.get(key, 0) + 1
    if _PLAN_BUILDS[key] == _PLAN_REBUILD_WARN:
        import warnings

        warnings.warn(f"pynqs_amd: the integral plan for sorb = {sorb} ({pl.buf.numel() * pl.buf.element_size() / 2**20:.0f} MiB) has been rebuilt "
                      f"{_PLAN_REBUILD_WARN} times: the caller passes new h1e / h2e tensor objects (or modifies them in place) on every call. "
                      "Keep the same tensors alive between calls to reuse the plan.", RuntimeWarning, stacklevel=3)
    return pl


_F64_COPIES: "list[tuple]" = []  # (weakref(h1e), weakref(h2e), versions, h1e as float64, h2e as float64), most recent first


def integrals_f64(h1e: Tensor, h2e: Tensor) -> Tuple[Tensor, Tensor]:
    """float64 copies of float32 integrals (cached per tensor pair like the plans), for the fused local-energy kernels, which
    exist in float64 only: every float32 value is a float64 value, so the kernels see exactly the caller's numbers and only the
    accumulation is done in (more than) the precision the reference's float32 path has (cpu_tensor.cpp:249,298 dispatch)."""
    import weakref

    if h1e.dtype == torch.float64:
        return h1e, h2e
    ver = (_ver(h1e), _ver(h2e), h1e.data_ptr(), h2e.data_ptr())
    trackable = ver[0] >= 0 and ver[1] >= 0
    for i, (r1, r2, v, a, b) in enumerate(_F64_COPIES if trackable else ()):
        if r1() is h1e and r2() is h2e and v == ver:
            if i:
                _F64_COPIES.insert(0, _F64_COPIES.pop(i))
            return a, b
    a, b = h1e.double(), h2e.double()
    if trackable:
        _F64_COPIES.insert(0, (weakref.ref(h1e), weakref.ref(h2e), ver, a, b))
        del _F64_COPIES[_MAX_PLANS:]
    return a, b


def get_Num_SinglesDoubles(sorb: int, noA: int, noB: int) -> int:
    """utils/public_function.py:132 / cpp_src/cpu/excitation.cpp:8-16."""
    return int(N.lib().pynqs_num_sd(sorb, noA, noB))


# ---- API -------------------------------------------------------------------------------------------
def check_sorb(sorb: int, nele: int) -> None:
    """bind.cpp:282-301.  The word count is dispatched at run time, so only sorb > 192 is a length error."""
    N.check(N.lib().pynqs_check_sorb(int(sorb), int(nele)), "check_sorb")


def tensor_to_onv(bra: Tensor, sorb: int) -> Tensor:
    """bind.cpp:9-22 -> cpu_tensor.cpp:8-44: 0/1 uint8 states (1-D or 2-D) -> packed onv uint8[n, 8*len]."""
    _contig(bra, "bra_tensor")
    if bra.dtype != torch.uint8 or bra.dim() not in (1, 2):
        raise RuntimeError("bra_tensor must be a 1-D or 2-D torch.uint8 tensor")
    L = _bra_len(sorb)
    if bra.numel() == 0:
        return torch.empty((0, 8 * L), dtype=torch.uint8, device=bra.device)
    if bra.numel() % sorb != 0:
        raise RuntimeError(f"shape {tuple(bra.shape)} is invalid for sorb = {sorb}")  # view(-1, sorb) fails
    dev, (x,), cpu = _stage(bra)
    x = x.view(-1, sorb)
    out = torch.empty((x.size(0), 8 * L), dtype=torch.uint8, device=dev)
    N.check(N.lib().pynqs_pm01_to_onv(x.data_ptr(), x.size(0), sorb, out.data_ptr(), _stream(dev)), "tensor_to_onv")
    return out.cpu() if cpu else out


def onv_to_tensor(bra: Tensor, sorb: int) -> Tensor:
    """bind.cpp:24-37 -> cpu_tensor.cpp:46-88: onv -> +-1 in torch.get_default_dtype(), shape [n, sorb]."""
    _check_onv(bra, "bra_tensor", sorb, (1, 2))
    dtype = torch.get_default_dtype()
    if dtype not in (torch.float32, torch.float64):
        raise RuntimeError("default dtype must be float32 or float64")
    if bra.numel() == 0:
        return torch.empty((0, sorb), dtype=dtype, device=bra.device)
    dev, (x,), cpu = _stage(bra)
    x = x.view(-1, x.size(-1))
    out = torch.empty((x.size(0), sorb), dtype=dtype, device=dev)
    code = N.PYNQS_F64 if dtype == torch.float64 else N.PYNQS_F32
    N.check(N.lib().pynqs_onv_to_pm1(x.data_ptr(), x.size(0), sorb, code, out.data_ptr(), _stream(dev)), "onv_to_tensor")
    return out.cpu() if cpu else out


# The reference's REDUCE and SAMPLE_SPACE local energies call get_comb_tensor(x) and then get_hij_torch(x, comb_x) on its result
# (vmc/energy/eloc.py:243-252,370-378) instead of the fused entry.  The last S+D list handed out is remembered by identity, and a
# get_hij_torch call on exactly that pair of tensor objects (unmodified: same storage, same version counters) is answered by the
# fused plan kernel in its Hmat-only form -- bit-identical values at a fifth of the generic pair kernel's time.
_last_comb = None  # (weakref(comb), comb ptr, comb version, weakref(bra), bra ptr, bra version, sorb, nele, noA, noB)
REUSE_COMB = True
CHECK_COMB_REUSE = __import__("os").environ.get("PYNQS_CHECK_COMB_REUSE", "0") == "1"


def _remember_comb(comb: Tensor, bra: Tensor, sorb: int, nele: int, noA: int, noB: int) -> None:
    global _last_comb
    import weakref

    _last_comb = (weakref.ref(comb), comb.data_ptr(), _ver(comb), weakref.ref(bra), bra.data_ptr(), _ver(bra), sorb, nele, noA, noB) \
        if comb.is_cuda and bra.is_cuda and _ver(comb) >= 0 and _ver(bra) >= 0 else None


def _is_last_comb(bra: Tensor, ket: Tensor, sorb: int, nele: int):
    """(noA, noB) if `ket` is the S+D list last produced for `bra` and neither has been written since, else None."""
    c = _last_comb
    if c is None or not REUSE_COMB or c[0]() is not ket or c[3]() is not bra:
        return None
    if (ket.data_ptr(), _ver(ket), bra.data_ptr(), _ver(bra), sorb, nele) != (c[1], c[2], c[4], c[5], c[6], c[7]):
        return None
    return c[8], c[9]


def get_comb_tensor(bra: Tensor, sorb: int, nele: int, noA: int, noB: int, flag_bit: bool = False) -> Tuple[Tensor, Tensor]:
    """bind.cpp:66-83 -> cpu_tensor.cpp:164-218.  Returns (comb uint8[n, ncomb, 8*len], states);
    states = +-1 double[n, ncomb, sorb] if flag_bit else torch.ones(1, float64) on the CPU (cpu_tensor.cpp:191)."""
    _check_onv(bra, "bra_tensor", sorb, (1, 2))
    L = _bra_len(sorb)
    ncomb = get_Num_SinglesDoubles(sorb, noA, noB) + 1
    x2 = bra.view(-1, 8 * L)
    n = x2.size(0) if bra.numel() else 0
    if n == 0:
        return (torch.empty((0, ncomb, 8 * L), dtype=torch.uint8, device=bra.device),
                torch.empty((0, ncomb, sorb), dtype=torch.float64, device=bra.device))
    dev, (x,), cpu = _stage(x2)
    comb = torch.empty((n, ncomb, 8 * L), dtype=torch.uint8, device=dev)
    pm = torch.empty((n, ncomb, sorb), dtype=torch.float64, device=dev) if flag_bit else None
    N.check(N.lib().pynqs_comb(x.data_ptr(), n, sorb, noA, noB, comb.data_ptr(), pm.data_ptr() if flag_bit else None,
                               _stream(dev)), "get_comb_tensor")
    if not flag_bit:
        pm = torch.ones(1, dtype=torch.float64)
    elif cpu:
        pm = pm.cpu()
    if not cpu and bra.dim() == 2:
        _remember_comb(comb, bra, sorb, nele, noA, noB)
    return (comb.cpu() if cpu else comb), pm


def get_comb_hij_fused(bra: Tensor, h1e: Tensor, h2e: Tensor, sorb: int, nele: int, noA: int, noB: int) -> Tuple[Tensor, Tensor]:
    """bind.cpp:239-250 -> cpu_tensor.cpp:220-272.  Returns (comb uint8[n, ncomb, 8*len], Hmat T[n, ncomb]),
    T = dtype of h1e; column 0 is x itself / <x|H|x>."""
    _check_onv(bra, "bra_tensor", sorb, (2,))
    code = _fdtype(h1e, h2e)
    _check_integrals(h1e, h2e, sorb)
    L = _bra_len(sorb)
    ncomb = get_Num_SinglesDoubles(sorb, noA, noB) + 1
    n = bra.size(0)
    if bra.numel() == 0:
        return (torch.empty((0, ncomb, 8 * L), dtype=torch.uint8, device=bra.device),
                torch.empty((0, ncomb), dtype=h1e.dtype, device=h1e.device))
    plan = plan_for(h1e, h2e, sorb, bra.device) if USE_PLAN else None
    if plan is not None:
        dev = plan.device
        x = bra if bra.device == dev else bra.to(dev)
        cpu = bra.device.type == "cpu" and h1e.device.type == "cpu" and h2e.device.type == "cpu"
        comb = torch.empty((n, ncomb, 8 * L), dtype=torch.uint8, device=dev)
        hmat = torch.empty((n, ncomb), dtype=h1e.dtype, device=dev)
        N.check(N.lib().pynqs_comb_hij_fused_plan(x.data_ptr(), n, sorb, nele, noA, noB, plan.data_ptr(), code,
                                                  comb.data_ptr(), hmat.data_ptr(), _stream(dev)), "get_comb_hij_fused")
        if not cpu and x is bra:
            _remember_comb(comb, bra, sorb, nele, noA, noB)  # (a second operator on the same list: get_hij_torch(x, comb, h1e_spin, ...))
        return (comb.cpu(), hmat.cpu()) if cpu else (comb, hmat)
    dev, (x, a, b), cpu = _stage(bra, h1e, h2e)
    comb = torch.empty((n, ncomb, 8 * L), dtype=torch.uint8, device=dev)
    hmat = torch.empty((n, ncomb), dtype=h1e.dtype, device=dev)
    N.check(N.lib().pynqs_comb_hij_fused(x.data_ptr(), n, sorb, nele, noA, noB, a.data_ptr(), b.data_ptr(), code,
                                         comb.data_ptr(), hmat.data_ptr(), _stream(dev)), "get_comb_hij_fused")
    return (comb.cpu(), hmat.cpu()) if cpu else (comb, hmat)


def get_hij_torch(bra: Tensor, ket: Tensor, h1e: Tensor, h2e: Tensor, sorb: int, nele: int) -> Tensor:
    """bind.cpp:39-64 -> cpu_tensor.cpp:274-325.  ket 3-D [n, m, .]: Hmat[i, j] = <bra_i|H|ket_ij>;
    ket 2-D [m, .]: the full matrix <bra_i|H|ket_j>.  Degree > 2 -> 0."""
    _check_onv(bra, "bra_tensor", sorb, (2,))
    _check_onv(ket, "ket_tensor", sorb, (2, 3))
    code = _fdtype(h1e, h2e)
    _check_integrals(h1e, h2e, sorb)
    n = bra.size(0)
    is3d = ket.dim() == 3
    m = ket.size(1) if is3d else ket.size(0)
    if is3d and ket.size(0) != n:
        raise RuntimeError(f"ket.size(0) = {ket.size(0)} != bra.size(0) = {n}")
    if bra.numel() == 0 or ket.numel() == 0:
        return torch.empty((n, m), dtype=h1e.dtype, device=h1e.device)
    same = _is_last_comb(bra, ket, sorb, nele) if is3d and USE_PLAN else None
    if same is not None and h1e.device == bra.device == h2e.device:
        plan = plan_for(h1e, h2e, sorb, bra.device)
        if plan is not None:
            hmat = torch.empty((n, m), dtype=h1e.dtype, device=bra.device)
            N.check(N.lib().pynqs_comb_hij_fused_plan(bra.data_ptr(), n, sorb, nele, same[0], same[1], plan.data_ptr(), code, None,
                                                      hmat.data_ptr(), _stream(bra.device)), "get_hij_torch")
            if CHECK_COMB_REUSE:
                # the shortcut trusts identity + version counters: with PYNQS_CHECK_COMB_REUSE=1 a few columns are recomputed from the
                # kets actually passed in (generic pair kernel) -- a caller that wrote into comb without bumping its version shows up here
                cols = torch.randint(0, m, (min(m, 8),), device=ket.device)
                sub = ket[:, cols].contiguous()
                chk = torch.empty((n, cols.numel()), dtype=h1e.dtype, device=bra.device)
                N.check(N.lib().pynqs_hij(bra.data_ptr(), n, sub.data_ptr(), cols.numel(), 1, h1e.data_ptr(), h2e.data_ptr(), code, sorb, nele,
                                          chk.data_ptr(), _stream(bra.device)), "get_hij_torch (check)")
                if not torch.equal(chk, hmat[:, cols]):
                    raise RuntimeError("get_hij_torch: `ket` is not the S+D list get_comb_tensor returned for `bra` any more (modified in place "
                                       "through a path that does not bump the version counter?)")
            return hmat
    dev, (x, k, a, b), cpu = _stage(bra, ket, h1e, h2e)
    hmat = torch.empty((n, m), dtype=h1e.dtype, device=dev)
    N.check(N.lib().pynqs_hij(x.data_ptr(), n, k.data_ptr(), m, int(is3d), a.data_ptr(), b.data_ptr(), code, sorb, nele,
                              hmat.data_ptr(), _stream(dev)), "get_hij_torch")
    return hmat.cpu() if cpu else hmat


def wavefunction_lut(bra_key: Tensor, onv: Tensor, sorb: int, little_endian: bool = True) -> Tuple[Tensor, Tensor]:
    """bind.cpp:216-236 -> cpu_tensor.cpp:642-688: binary search of onv in the sorted bra_key.
    Returns (idx int64[n] with -1 for misses, mask bool[n]).  If either input is on the CPU the result
    is on the CPU, as in the reference."""
    _check_onv(bra_key, "bra_key", sorb, (2,))
    _check_onv(onv, "onv", sorb, (2,))
    if not little_endian:
        # keys sorted as BIG-endian multi-word integers (most significant word first; the reference's own branch for this order
        # decrements its loop index the wrong way, cpu_tensor.cpp:613, and is used nowhere): the same search on word-reversed copies
        L = _bra_len(sorb)
        rev = lambda t: t.view(-1, L, 8).flip(1).reshape(-1, 8 * L).contiguous()  # noqa: E731
        return wavefunction_lut(rev(bra_key), rev(onv), sorb, True)
    n = onv.size(0)
    any_cpu = bra_key.device.type == "cpu" or onv.device.type == "cpu"
    if onv.numel() == 0:
        d = torch.device("cpu") if any_cpu else onv.device
        return torch.zeros(0, dtype=torch.int64, device=d), torch.zeros(0, dtype=torch.bool, device=d)
    dev, (k, q), _ = _stage(bra_key, onv)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    mask = torch.empty(n, dtype=torch.bool, device=dev)
    N.check(N.lib().pynqs_wavefunction_lut(k.data_ptr(), k.size(0), q.data_ptr(), n, sorb, idx.data_ptr(), mask.data_ptr(),
                                           _stream(dev)), "wavefunction_lut")
    return (idx.cpu(), mask.cpu()) if any_cpu else (idx, mask)


class HashTable:
    """Device hash table over a set of onv keys (the reference's optional `HashTable`, bind.cpp:363-379 /
    cuda/hashTable.cu).  Values are positions in the key array handed to hash_build."""

    def __init__(self, table: Tensor, nkeys: int, sorb: int) -> None:
        self.table, self.nkeys, self.sorb = table, nkeys, sorb

    @property
    def memory(self) -> int:
        return self.table.numel() * self.table.element_size()

    # the reference's table is an array of buckets (cuda/hashTable.cu); this one is open addressing with one key per slot:
    # bucketNum = number of slots (a power of two >= 4 * nkeys), bucketSize = 1
    @property
    def bucketNum(self) -> int:
        cap = 64
        while cap < 4 * max(self.nkeys, 1):
            cap *= 2
        return cap

    @property
    def bucketSize(self) -> int:
        return 1

    @staticmethod
    def bitWidth() -> int:
        """sizeof(key word) in bytes, as the reference's HashTable.bitWidth (libs/C_extension.pyi:385-389)."""
        return 8

    def cleanMemory(self) -> None:
        self.table = None


def hash_build(bra_key: Tensor, sorb: int) -> HashTable:
    """cuda_tensor.cpp:489-534 (hash_build): keys uint8[nkeys, 8*len] (distinct) -> HashTable on the GPU."""
    _check_onv(bra_key, "bra_key", sorb, (2,))
    dev, (k,), _ = _stage(bra_key)
    nbytes = N.lib().pynqs_hash_bytes(k.size(0), sorb)
    table = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
    N.check(N.lib().pynqs_hash_build(k.data_ptr(), k.size(0), sorb, table.data_ptr(), _stream(dev)), "hash_build")
    if k is not bra_key:
        torch.cuda.current_stream(dev).synchronize()
    return HashTable(table, k.size(0), sorb)


class KeysIndex:
    """Block index of a key table for the INDEXED key-major SAMPLE_SPACE kernel (include/pynqs_amd.h: pynqs_keys_index_build): per block
    of the orbitals the keys' block values sorted, with the key numbers.  `per_walker`: keys a table member meets through the index
    (the streamed form meets nkeys)."""

    def __init__(self, index: Tensor, nkeys: int, sorb: int, per_walker: float) -> None:
        self.index, self.nkeys, self.sorb, self.per_walker = index, nkeys, sorb, per_walker

    @property
    def memory(self) -> int:
        return self.index.numel() * self.index.element_size()


def keys_index_build(bra_key: Tensor, sorb: int) -> KeysIndex:
    """keys uint8[nkeys, 8*len] (distinct, any order, on the GPU) -> KeysIndex.  One host synchronisation (the density read-back)."""
    _check_onv(bra_key, "bra_key", sorb, (2,))
    if not bra_key.is_cuda:
        raise ValueError("keys_index_build: the keys must be on the GPU")
    k = bra_key.contiguous()
    dev, nk, lib = k.device, k.size(0), N.lib()
    nbytes, wbytes = lib.pynqs_keys_index_bytes(nk, sorb), lib.pynqs_keys_index_workspace(nk, sorb)
    if nbytes < 0 or wbytes < 0:
        raise ValueError(f"keys_index_build: sorb must be even and nkeys < 2^27 (sorb {sorb}, {nk} keys)")
    index = torch.empty(max((nbytes + 7) // 8, 1), dtype=torch.int64, device=dev)
    work = torch.empty(max(wbytes, 8), dtype=torch.uint8, device=dev)
    total = torch.empty(1, dtype=torch.int64, device=dev)
    N.check(lib.pynqs_keys_index_build(k.data_ptr(), nk, sorb, index.data_ptr(), work.data_ptr(), _stream(dev)), "keys_index_build")
    N.check(lib.pynqs_keys_index_density(index.data_ptr(), nk, sorb, total.data_ptr(), _stream(dev)), "keys_index_density")
    return KeysIndex(index, nk, sorb, float(total.item()) / max(nk, 1))


def hash_lookup(ht: HashTable, onv: Tensor) -> Tuple[Tensor, Tensor]:
    """cuda_tensor.cpp:536-559 (hash_lookup): (idx int64[n] or -1, mask bool[n]); same answers as wavefunction_lut
    on the key array the table was built from."""
    _check_onv(onv, "onv", ht.sorb, (2,))
    n = onv.size(0)
    dev = ht.table.device
    q = onv if onv.device == dev else onv.to(dev)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    mask = torch.empty(n, dtype=torch.bool, device=dev)
    if n:
        N.check(N.lib().pynqs_hash_lookup(ht.table.data_ptr(), ht.nkeys, q.data_ptr(), n, ht.sorb, idx.data_ptr(), mask.data_ptr(),
                                          _stream(dev)), "hash_lookup")
    return (idx.cpu(), mask.cpu()) if onv.device.type == "cpu" else (idx, mask)


_SPIN_FLIP_CALLS = 0


class RBMTable:
    """Device-resident re-layout of a real RBM's parameters for the fused SIMPLE local energy
    (include/pynqs_amd.h: pynqs_rbm_table_build; reference amplitude: vmc/ansatz/rbm/rbm.py:186-211).
    weights [num_hidden, sorb], hidden_bias [num_hidden], visible_bias [sorb] or None; float64.
    Rebuild after every parameter update (one small kernel)."""

    def __init__(self, weights: Tensor, hidden_bias: Tensor, visible_bias: "Tensor | None" = None) -> None:
        if weights.dim() != 2 or hidden_bias.numel() != weights.size(0):
            raise RuntimeError("weights must be [num_hidden, sorb] and hidden_bias [num_hidden]")
        if visible_bias is not None and visible_bias.numel() != weights.size(1):
            raise RuntimeError("visible_bias must be [sorb]")
        ts = [weights, hidden_bias] + ([visible_bias] if visible_bias is not None else [])
        if any(t.dtype != torch.float64 for t in ts):
            raise RuntimeError("the fused RBM local energy is float64 only")
        src = ts
        dev, ts, _ = _stage(*[t.detach().contiguous() for t in ts])
        self.nhidden, self.sorb, self.device = int(weights.size(0)), int(weights.size(1)), dev
        nbytes = N.lib().pynqs_rbm_table_bytes(self.sorb, self.nhidden)
        if nbytes < 0:
            raise RuntimeError(f"bad RBM sizes: sorb = {self.sorb}, num_hidden = {self.nhidden}")
        self.buf = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
        N.check(N.lib().pynqs_rbm_table_build(ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr() if len(ts) > 2 else None,
                                              self.sorb, self.nhidden, self.buf.data_ptr(), _stream(dev)), "rbm_table_build")
        if any(a.data_ptr() != b.data_ptr() for a, b in zip(src, ts)):
            torch.cuda.current_stream(dev).synchronize()  # staging copies must outlive the build kernel

    def data_ptr(self) -> int:
        return self.buf.data_ptr()


RBM_FLAVOURS = {"real": N.RBM_REAL, "tanh": N.RBM_TANH, "pRBM": N.RBM_PHASE}  # rbm_type (rbm.py:199-211) -> include/pynqs_amd.h


def eloc_rbm(bra: Tensor, h1e: Tensor, h2e: Tensor, table: RBMTable, sorb: int, nele: int, noA: int, noB: int,
             want_psi: bool = True, rbm_type: str = "real") -> Tuple[Tensor, "Tensor | None"]:
    """SIMPLE local energy with the RBM amplitude ratio evaluated on chip (pynqs_eloc_rbm_flavour):
    (eloc[n], psi(x)[n] or None), float64 for rbm_type "real" / "tanh", complex128 for "pRBM".  Equivalent to
    vmc/energy/eloc.py:121-203 with ansatz = RBMWavefunction(rbm_type=...)."""
    _check_onv(bra, "bra", sorb, (2,))
    if rbm_type not in RBM_FLAVOURS:
        raise RuntimeError(f"rbm_type {rbm_type!r} has no fused local energy (fused: {sorted(RBM_FLAVOURS)})")
    if table.sorb != sorb:
        raise RuntimeError(f"RBM table was built for sorb = {table.sorb}, not {sorb}")
    if _fdtype(h1e, h2e) != N.PYNQS_F64:
        raise RuntimeError("the fused RBM local energy is float64 only")
    plan = plan_for(h1e, h2e, sorb, bra.device)
    if plan is None:
        raise RuntimeError("the fused RBM local energy needs an even sorb")
    dev, (x,), all_cpu = _stage(bra)
    if plan.device != dev or table.device != dev:
        raise RuntimeError("bra, integrals and RBM table must be on the same device")
    n = x.size(0)
    dt = torch.complex128 if rbm_type == "pRBM" else torch.float64
    eloc = torch.empty(n, dtype=dt, device=dev)
    psi = torch.empty(n, dtype=dt, device=dev) if want_psi else None
    N.check(N.lib().pynqs_eloc_rbm_flavour(x.data_ptr(), n, sorb, nele, noA, noB, plan.data_ptr(), table.data_ptr(), table.nhidden,
                                           RBM_FLAVOURS[rbm_type], eloc.data_ptr(), psi.data_ptr() if want_psi else None, _stream(dev)),
            "pynqs_eloc_rbm")
    if all_cpu and bra.device.type == "cpu":
        return eloc.cpu(), (psi.cpu() if want_psi else None)
    return eloc, psi


def rbm_forward(onv: Tensor, weights: Tensor, hidden_bias: Tensor, visible_bias: "Tensor | None", sorb: int, rbm_type: str = "real") -> Tensor:
    """psi(x) of the reference's RBM amplitudes (vmc/ansatz/rbm/rbm.py:186-211) on a list of determinants, one kernel
    (pynqs_rbm_forward): onv uint8[n, 8 len] -> psi float64[n] (rbm_type "real" / "tanh") or complex128[n] ("pRBM"; "complex" with
    weights [H, sorb, 2], hidden_bias [H, 2], visible_bias [sorb, 2] as (re, im) pairs: the reference's params_* layout)."""
    _check_onv(onv, "onv", sorb, (2,))
    flav = {"real": N.RBM_REAL, "tanh": N.RBM_TANH, "pRBM": N.RBM_PHASE, "complex": N.RBM_COMPLEX}.get(rbm_type)
    if flav is None:
        raise RuntimeError(f"rbm_type {rbm_type!r} has no fused forward")
    cplx_par = rbm_type == "complex"
    W = weights.detach().double().contiguous()
    hb = hidden_bias.detach().double().contiguous()
    vb = visible_bias.detach().double().contiguous() if visible_bias is not None else None
    H = W.size(0)
    if W.shape[:2] != (H, sorb) or W.dim() != (3 if cplx_par else 2) or hb.numel() != H * (2 if cplx_par else 1) or \
            (vb is not None and vb.numel() != sorb * (2 if cplx_par else 1)):
        raise RuntimeError("RBM parameter shapes do not match sorb / num_hidden")
    if not (onv.is_cuda and W.is_cuda and hb.is_cuda and (vb is None or vb.is_cuda)):
        raise RuntimeError("rbm_forward: determinants and parameters must be on the GPU")
    n = onv.size(0)
    out_c = rbm_type in ("pRBM", "complex")
    psi = torch.empty(n, dtype=torch.complex128 if out_c else torch.float64, device=onv.device)
    N.check(N.lib().pynqs_rbm_forward(onv.data_ptr(), n, sorb, W.data_ptr(), hb.data_ptr(), vb.data_ptr() if vb is not None else None, H, flav,
                                      psi.data_ptr(), _stream(onv.device)), "pynqs_rbm_forward")
    return psi


def _rbm_params(weights, hidden_bias, visible_bias, sorb, rbm_type):
    flav = {"real": N.RBM_REAL, "tanh": N.RBM_TANH, "pRBM": N.RBM_PHASE, "complex": N.RBM_COMPLEX}.get(rbm_type)
    if flav is None:
        raise RuntimeError(f"rbm_type {rbm_type!r} has no fused forward")
    cplx_par = rbm_type == "complex"
    W = weights.detach().double().contiguous()
    hb = hidden_bias.detach().double().contiguous()
    vb = visible_bias.detach().double().contiguous() if visible_bias is not None else None
    H = W.size(0)
    if W.shape[:2] != (H, sorb) or W.dim() != (3 if cplx_par else 2) or hb.numel() != H * (2 if cplx_par else 1) or \
            (vb is not None and vb.numel() != sorb * (2 if cplx_par else 1)):
        raise RuntimeError("RBM parameter shapes do not match sorb / num_hidden")
    if not (W.is_cuda and hb.is_cuda and (vb is None or vb.is_cuda)):
        raise RuntimeError("RBM parameters must be on the GPU")
    return flav, W, hb, vb, H


def rbm_forward_children_supported(sorb: int, num_hidden: int, rbm_type: str = "real") -> bool:
    flav = {"real": N.RBM_REAL, "tanh": N.RBM_TANH, "pRBM": N.RBM_PHASE, "complex": N.RBM_COMPLEX}.get(rbm_type)
    return flav is not None and bool(N.lib().pynqs_rbm_forward_children_supported(sorb, num_hidden, flav))


def rbm_forward_children(onv: Tensor, parent: Tensor, walkers: Tensor, weights: Tensor, hidden_bias: Tensor, visible_bias: "Tensor | None",
                         sorb: int, rbm_type: str = "real", count: "Tensor | None" = None, out: "Tensor | None" = None) -> Tensor:
    """psi of the reference's RBM amplitudes (vmc/ansatz/rbm/rbm.py:186-211) on the DISTINCT x' of a REDUCE front end, each from its parent
    walker: onv[r] is walkers[parent[r]] with at most four orbitals flipped (ReduceFrontEnd.uniq_onv / .uniq_parent), so the factors
    1 + exp(-2 theta_h(x')) follow from the walker's by 4 table multiplications per hidden unit -- no exponential, no loop over the orbitals
    (pynqs_rbm_children_prepare on the walkers + pynqs_rbm_forward_children).  count: int32 device tensor whose
    first element is the number of valid rows (the front end's counters; rows past it are not computed), or None for all rows."""
    _check_onv(onv, "onv", sorb, (2,))
    _check_onv(walkers, "walkers", sorb, (2,))
    flav, W, hb, vb, H = _rbm_params(weights, hidden_bias, visible_bias, sorb, rbm_type)
    dev = onv.device
    if not (onv.is_cuda and walkers.device == dev and parent.device == dev and parent.dtype == torch.int32 and parent.numel() >= onv.size(0)):
        raise RuntimeError("rbm_forward_children: onv, walkers and parent (int32) on one GPU")
    n, nw = onv.size(0), walkers.size(0)
    nbytes = N.lib().pynqs_rbm_children_table_bytes(nw, sorb, H, flav)
    table = torch.empty(max(nbytes // 8, 1), dtype=torch.float64, device=dev)
    st = _stream(dev)
    wk = walkers.contiguous()
    N.check(N.lib().pynqs_rbm_children_prepare(wk.data_ptr(), nw, sorb, W.data_ptr(), hb.data_ptr(), vb.data_ptr() if vb is not None else None,
                                               H, flav, table.data_ptr(), st), "pynqs_rbm_children_prepare")
    out_c = rbm_type in ("pRBM", "complex")
    psi = out if out is not None else torch.empty(n, dtype=torch.complex128 if out_c else torch.float64, device=dev)
    N.check(N.lib().pynqs_rbm_forward_children(onv.contiguous().data_ptr(), n, count.data_ptr() if count is not None else None, parent.data_ptr(),
                                               wk.data_ptr(), nw, table.data_ptr(), sorb, W.data_ptr(), hb.data_ptr(),
                                               vb.data_ptr() if vb is not None else None, H, flav, psi.data_ptr(), st), "pynqs_rbm_forward_children")
    return psi


class CRBMTable:
    """Device-resident re-layout of an RBM with COMPLEX parameters for the fused SIMPLE local energy (pynqs_crbm_table_build;
    rbm.py:199-211, rbm_type "complex").  weights [num_hidden, sorb], hidden_bias [num_hidden], visible_bias [sorb] or None:
    complex128 tensors, or float64 tensors with a trailing (re, im) axis (the reference's params_* layout)."""

    def __init__(self, weights: Tensor, hidden_bias: Tensor, visible_bias: "Tensor | None" = None) -> None:
        def as_pairs(t: Tensor) -> Tensor:
            t = t.detach()
            if t.is_complex():
                t = torch.view_as_real(t.to(torch.complex128))
            if t.dtype != torch.float64 or t.size(-1) != 2:
                raise RuntimeError("complex RBM parameters must be complex128 or float64 (re, im) pairs")
            return t.contiguous()

        w, hb = as_pairs(weights), as_pairs(hidden_bias).reshape(-1, 2)
        vb = as_pairs(visible_bias).reshape(-1, 2) if visible_bias is not None else None
        if w.dim() != 3 or hb.size(0) != w.size(0) or (vb is not None and vb.size(0) != w.size(1)):
            raise RuntimeError("weights must be [num_hidden, sorb], hidden_bias [num_hidden], visible_bias [sorb]")
        ts = [w, hb] + ([vb] if vb is not None else [])
        dev, st, _ = _stage(*ts)
        self.nhidden, self.sorb, self.device = int(w.size(0)), int(w.size(1)), dev
        nbytes = N.lib().pynqs_crbm_table_bytes(self.sorb, self.nhidden)
        if nbytes < 0:
            raise RuntimeError(f"bad RBM sizes: sorb = {self.sorb}, num_hidden = {self.nhidden}")
        self.buf = torch.empty(nbytes // 8, dtype=torch.float64, device=dev)
        N.check(N.lib().pynqs_crbm_table_build(st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr() if len(st) > 2 else None,
                                               self.sorb, self.nhidden, self.buf.data_ptr(), _stream(dev)), "crbm_table_build")
        if any(a.device != dev for a in ts):
            torch.cuda.current_stream(dev).synchronize()  # staging copies of host parameters must outlive the build kernel
        # (device-side temporaries -- contiguous / view_as_real copies -- are freed in stream order: the caching allocator hands their
        # memory only to later work on the same stream)

    def data_ptr(self) -> int:
        return self.buf.data_ptr()


def eloc_crbm(bra: Tensor, h1e: Tensor, h2e: Tensor, table: CRBMTable, sorb: int, nele: int, noA: int, noB: int,
              want_psi: bool = True, log_scale: float = 0.0) -> Tuple[Tensor, "Tensor | None"]:
    """SIMPLE local energy with the amplitude ratio of a complex-parameter RBM evaluated on chip (pynqs_eloc_crbm):
    (eloc complex128[n], psi(x) exp(-log_scale) complex128[n] or None)."""
    _check_onv(bra, "bra", sorb, (2,))
    if table.sorb != sorb:
        raise RuntimeError(f"RBM table was built for sorb = {table.sorb}, not {sorb}")
    if _fdtype(h1e, h2e) != N.PYNQS_F64:
        raise RuntimeError("the fused RBM local energy is float64 only")
    plan = plan_for(h1e, h2e, sorb, bra.device)
    if plan is None:
        raise RuntimeError("the fused RBM local energy needs an even sorb")
    dev, (x,), all_cpu = _stage(bra)
    if plan.device != dev or table.device != dev:
        raise RuntimeError("bra, integrals and RBM table must be on the same device")
    n = x.size(0)
    eloc = torch.empty(n, dtype=torch.complex128, device=dev)
    psi = torch.empty(n, dtype=torch.complex128, device=dev) if want_psi else None
    N.check(N.lib().pynqs_eloc_crbm(x.data_ptr(), n, sorb, nele, noA, noB, plan.data_ptr(), table.data_ptr(), table.nhidden, float(log_scale),
                                    eloc.data_ptr(), psi.data_ptr() if want_psi else None, _stream(dev)), "pynqs_eloc_crbm")
    if all_cpu and bra.device.type == "cpu":
        return eloc.cpu(), (psi.cpu() if want_psi else None)
    return eloc, psi


# ---- two ansatz-side helpers of the reference module (outside the local-energy path) so that its autoregressive
# ansaetze (BDG-RNN / MPS-RNN: permute_sgn; symmetry masks of the AR samplers: constrain_make_charts) import AND run on the drop-in.
# Plain tensor algebra on the inputs' device: they are called once per forward on [nbatch, sorb] inputs.
_CHARTS = None


def constrain_make_charts(sym_idex: Tensor) -> Tensor:
    """cpu_tensor.cpp:558-588: 9-entry lookup of the symmetry-constraint charts, double[nbatch, 4] (the parameter name is the
    reference's, libs/C_extension.pyi:278); indices outside {10, 6, 14, 9, 5, 13, 11, 7, 15} give zeros (the reference reads a
    default-constructed map entry there)."""
    sym_index = sym_idex
    global _CHARTS
    if _CHARTS is None:
        t = torch.zeros((16, 4), dtype=torch.float64)
        for k, row in zip((10, 6, 14, 9, 5, 13, 11, 7, 15), ((1, 0, 0, 0), (0, 0, 1, 0), (1, 0, 1, 0), (0, 1, 0, 0), (0, 0, 0, 1), (0, 1, 0, 1),
                                                             (1, 1, 0, 0), (0, 0, 1, 1), (1, 1, 1, 1))):
            t[k] = torch.tensor(row, dtype=torch.float64)
        _CHARTS = t
    idx = sym_index.reshape(-1).long()
    inside = (idx >= 0) & (idx < 16)
    out = _CHARTS.to(idx.device)[idx.clamp(0, 15)]
    return torch.where(inside.unsqueeze(1), out, torch.zeros_like(out))


def permute_sgn(image2: Tensor, onstate: Tensor, sorb: int) -> Tensor:
    """cpu_tensor.cpp:356-380 -> onstate.cpp:195-226: fermionic sign of re-ordering the orbitals into the sequence `image2` for the
    occupation vectors onstate[nbatch, sorb] (0 / 1): (-1)^(number of pairs of occupied orbitals whose order image2 inverts) --
    the reference counts the same pairs while it moves the orbitals one by one.  Returns double[nbatch] of +1 / -1."""
    if image2.numel() != sorb and onstate.size(-1) != sorb:
        raise ValueError("Dim error")  # std::length_error
    if onstate.size(0) == 0:
        return torch.zeros(0, dtype=torch.float64, device=onstate.device)
    order = image2.reshape(-1).long().to(onstate.device)
    occ = (onstate.reshape(onstate.size(0), -1)[:, order] != 0).to(torch.float64)  # occupations in the new order
    pos = torch.arange(order.numel(), device=order.device)
    inverted = ((pos.unsqueeze(1) < pos.unsqueeze(0)) & (order.unsqueeze(1) > order.unsqueeze(0))).to(torch.float64)  # [p, q]: p before q, labels reversed
    pairs = ((occ @ inverted) * occ).sum(1)
    return (1 - 2 * (pairs.long() & 1)).to(torch.float64)


def spin_flip_rand(bra: Tensor, sorb: int, nele: int, noA: int, noB: int, seed: int, in_place: bool = False) -> Tuple[Tensor, Tensor]:
    """bind.cpp:303-314 -> cpu_tensor.cpp:90-137: one random single/double move (or none) per walker.
    Returns (onv_to_tensor(new walkers), new walkers uint8[n, 8*len]).  The reference's generators are
    function-local statics seeded once, so its `seed` only matters on the first call; here
    every call uses (seed, a per-process call counter) so that successive calls give fresh, reproducible moves."""
    global _SPIN_FLIP_CALLS
    _check_onv(bra, "bra_tensor", sorb, (1, 2))
    x2 = bra.view(-1, bra.size(-1))
    n = x2.size(0)
    dev, (x,), cpu = _stage(x2)
    out = x if (in_place and not cpu) else torch.empty_like(x)
    if n:
        N.check(N.lib().pynqs_spin_flip_rand(x.data_ptr(), n, sorb, noA, noB, int(seed) & (2**64 - 1), _SPIN_FLIP_CALLS << 32,
                                             out.data_ptr(), _stream(dev)), "spin_flip_rand")
    _SPIN_FLIP_CALLS += 1
    if cpu:
        out = out.cpu()
        if in_place:
            x2.copy_(out)
            out = x2
    return onv_to_tensor(out, sorb), out


def merge_rank_sample(idx: Tensor, counts: Tensor, split_idx: Tensor, length: int) -> Tensor:
    """bind.cpp:190-201 -> cpu_tensor.cpp:537-556: merge_counts[idx[i]] += counts[i] (int64[length]); used
    after the cross-rank torch.unique in vmc/sample.py:675-685.  `split_idx` only steers the reference's
    race-avoiding launch schedule (cuda_tensor.cpp:403-411) and is ignored: index_add_ is atomic."""
    _contig(idx, "idx"); _contig(counts, "counts")
    out = torch.zeros(int(length), dtype=torch.int64, device=idx.device)
    return out.index_add_(0, idx.to(torch.int64), counts.to(torch.int64))


# ---- integral layout (host side, numpy in / numpy out like cpp_src/tensor/integral.cpp) -----------------
def _pair_maps(sorb: int):
    i, j = np.meshgrid(np.arange(sorb), np.arange(sorb), indexing="ij")
    hi, lo = np.maximum(i, j), np.minimum(i, j)
    pair = hi * (hi - 1) // 2 + lo  # meaningless on the diagonal (masked by the callers)
    sgn = np.where(i > j, 1.0, -1.0)
    return pair, sgn, i != j


def compress_h1e_h2e(h1e: np.ndarray, h2e: np.ndarray, sorb: int):
    """integral.cpp:6-60: h1e[s, s], antisymmetrised h2e[s, s, s, s] -> (h1e[s*s], h2e[pair(pair+1)/2]).
    When several (i,j,k,l) share a packed slot the LAST one in lexicographic order wins, as in the reference's loop: the blocks
    h2e[i] are visited in order and numpy's fancy assignment keeps the last write inside a block.  Working memory is one [s, s, s]
    block (the reference is an O(1)-memory loop; the input itself is s^4: 1.7 GB at sorb 120, 9.2 GB at sorb 184)."""
    h1e = np.asarray(h1e, dtype=np.float64)
    h2e = np.asarray(h2e, dtype=np.float64)
    if h1e.shape != (sorb, sorb) or h2e.shape != (sorb,) * 4:
        raise ValueError(f"expected h1e {(sorb, sorb)} and h2e {(sorb,) * 4}, got {h1e.shape} and {h2e.shape}")
    npair = sorb * (sorb - 1) // 2
    pair, sgn, off = _pair_maps(sorb)
    out = np.zeros(npair * (npair + 1) // 2, dtype=np.float64)
    kl, skl, okl = pair[None, :, :], sgn[None, :, :], off[None, :, :]
    for i in range(sorb):
        ij = pair[i][:, None, None]
        valid = off[i][:, None, None] & okl                 # [s, s, s] over (j, k, l)
        P, Q = np.maximum(ij, kl), np.minimum(ij, kl)
        out[(P * (P + 1) // 2 + Q)[valid]] = (sgn[i][:, None, None] * skl * h2e[i])[valid]
    return h1e.reshape(-1).copy(), out


def decompress_h1e_h2e(h1e: np.ndarray, h2e: np.ndarray, sorb: int):
    """integral.cpp:62-125; raises ValueError on a size mismatch (std::invalid_argument).  Elements with i == j or k == l are 0
    (the reference leaves them unwritten).  Working memory beyond the s^4 result: one [s, s, s] block."""
    h1e = np.asarray(h1e, dtype=np.float64).reshape(-1)
    h2e = np.asarray(h2e, dtype=np.float64).reshape(-1)
    npair = sorb * (sorb - 1) // 2
    if h1e.size != sorb * sorb:
        raise ValueError(f"h1e array size is incorrect: expected {sorb * sorb}, got {h1e.size}")
    if h2e.size != npair * (npair + 1) // 2:
        raise ValueError(f"h2e array size is incorrect: expected {npair * (npair + 1) // 2}, got {h2e.size}")
    pair, sgn, off = _pair_maps(sorb)
    full = np.zeros((sorb,) * 4, dtype=np.float64)
    kl, skl, okl = pair[None, :, :], sgn[None, :, :], off[None, :, :]
    for i in range(sorb):
        ij = pair[i][:, None, None]
        valid = off[i][:, None, None] & okl
        P, Q = np.maximum(ij, kl), np.minimum(ij, kl)
        slot = np.where(valid, P * (P + 1) // 2 + Q, 0)
        full[i] = np.where(valid, h2e[slot] * (sgn[i][:, None, None] * skl), 0.0)
    return h1e.reshape(sorb, sorb).copy(), full

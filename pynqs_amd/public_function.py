"""Host-side helpers of the local-energy path (mirror of the parts of PyNQS' utils/public_function.py that
sit on the path: SURVEY.md 8a rows a3, a16).  Same names and argument meaning as the reference so that
`vmc/energy`-style code reads the same; all determinant arithmetic goes through pynqs_amd.C_extension.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import C_extension as CX
from .C_extension import get_Num_SinglesDoubles, onv_to_tensor, wavefunction_lut  # noqa: F401 (re-exported)
from .distributed import get_rank, get_world_size

USE_HASH = True  # WavefunctionLUT keeps a GPU hash table next to the sorted keys (reference: USE_HASH = False, :23)


def check_para(bra: Tensor) -> None:
    """utils/public_function.py: onv tensors must be uint8."""
    if bra.dtype is not torch.uint8:
        raise Exception(f"The type of bra {bra.dtype} must be torch.uint8")


def split_batch_idx(dim: int, min_batch: int) -> List[int]:
    """utils/public_function.py:695-717: cumulative ends of chunks of `min_batch` (last one shorter)."""
    nchunks = -(-dim // min_batch)  # ceil
    return [min((c + 1) * min_batch, dim) for c in range(nchunks)]


def split_length_idx(dim: int, length: int) -> List[int]:
    """utils/public_function.py:720-746: `length` nearly equal parts, the first dim % length one longer."""
    base, extra = divmod(dim, length)
    ends, stop = [], 0
    for part in range(length):
        stop += base + (part < extra)
        ends.append(stop)
    return ends


def torch_lexsort(keys: List[Tensor], dim: int = -1) -> Tensor:
    """utils/public_function.py:615-648 (np.lexsort semantics: last key is the primary one): stable sorts from the
    least significant key upwards, each applied to the order found so far."""
    if len(keys) < 2:
        raise ValueError(f"keys must be at least 2 sequences, but {len(keys)=}.")
    order = torch.argsort(keys[0], dim=dim, stable=True)
    for key in keys[1:]:
        step = torch.argsort(torch.gather(key, dim, order), dim=dim, stable=True)
        order = torch.gather(order, dim, step)
    return order


def torch_sort_onv(bra: Tensor, little_endian: bool = True) -> Tensor:
    """utils/public_function.py:651-692: argsort of onv rows as little-endian big integers.
    The reference lexsorts byte columns; sorting the 64-bit words (most significant last) is the same order
    and 8x fewer passes."""
    if bra.dim() != 2:
        raise AssertionError("onv batch must be 2-D")
    if not little_endian:
        raise NotImplementedError("Little_endian has not been implemented")
    words = bra.contiguous().view(torch.int64)  # [n, len]; compare as unsigned
    perm = torch.arange(words.size(0), device=bra.device)
    sign = torch.iinfo(torch.int64).min
    for w in range(words.size(1)):  # least significant word first, stable sorts
        key = words[perm, w] ^ sign  # unsigned order of int64: flip the sign bit
        perm = perm[torch.argsort(key, stable=True)]
    return perm


def unique_onv(x: Tensor) -> Tuple[Tensor, Tensor]:
    """(unique rows, inverse) of a uint8 onv batch, like torch.unique(x, dim=0, return_inverse=True) up to the
    ORDER of the unique rows (callers only use rows[inverse]; vmc/energy/flip.py:44-50).  torch's row-wise unique is a
    multi-pass sort; on the GPU the rows go through a hash table of row indices instead (pynqs_unique_first: every
    row learns the first row with its determinant), and the unique rows come in order of first appearance:
    0.27 -> 0.09 ms on the 7e5 kept x' of 8192 Fe2S2 walkers.  CPU tensors: word-wise sorts."""
    assert x.dim() == 2 and x.dtype == torch.uint8 and x.size(1) % 8 == 0
    n, L = x.size(0), x.size(1) // 8
    if x.is_cuda and 0 < n < 2**30 and L <= 3:
        from . import _native as N

        xc = x.contiguous()
        dev = x.device
        ws = torch.empty(N.lib().pynqs_unique_workspace(n), dtype=torch.uint8, device=dev)
        first = torch.empty(n, dtype=torch.int32, device=dev)
        # any sorb with this word count gives the same kernel
        N.check(N.lib().pynqs_unique_first(xc.data_ptr(), n, 64 * L, ws.data_ptr(), first.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                "pynqs_unique_first")
        first = first.long()
        is_rep = first == torch.arange(n, device=dev)
        uid = torch.cumsum(is_rep, 0) - 1
        return xc[is_rep], uid[first]
    words = x.contiguous().view(torch.int64)  # [n, L]
    if L == 1 or n == 0:
        u, inv = torch.unique(words.view(-1) if L == 1 else words, return_inverse=True, dim=0 if L > 1 else None)
        return u.view(-1, L).view(torch.uint8).view(-1, 8 * L), inv
    order = torch_sort_onv(x)
    sw = words[order]
    new = torch.ones(n, dtype=torch.bool, device=x.device)
    new[1:] = (sw[1:] != sw[:-1]).any(dim=1)
    group = torch.cumsum(new, 0) - 1
    inv = torch.empty(n, dtype=torch.int64, device=x.device)
    inv[order] = group
    return sw[new].view(torch.uint8).view(-1, 8 * L), inv


class WavefunctionLUT:
    """utils/public_function.py:749-868: (onv -> psi) table over sorted keys.  Same public surface as the reference's
    class (bra_key, wf_value, dtype, memory, lookup, index_value, rank_begin / rank_end ...); look-ups run on the GPU
    through the hash table built next to the keys (USE_HASH) or the binary search of pynqs_amd.C_extension."""

    def __init__(self, bra_key: Tensor, wf_value: Tensor, sorb: int, device=None, sort: bool = True) -> None:
        check_para(bra_key)
        if bra_key.size(0) != wf_value.size(0):
            raise AssertionError("one amplitude per key")
        self.sort, self.sorb = sort, sorb
        keys, vals = bra_key, wf_value
        if sort:
            order = torch_sort_onv(bra_key)
            keys, vals = bra_key[order], wf_value[order]
            self.idx_sorted = torch.argsort(order, stable=True)  # position of the i-th input key in the sorted table
        self._bra_key = keys.to(device).contiguous()
        self._wf_value = vals.to(device)
        self.hashtable = None
        self._rebuild_hash()
        # the contiguous shard of the (unsorted) keys that belongs to this rank
        self.rank, self.world_size = get_rank(), get_world_size()
        self.rank_idx = [0] + split_length_idx(bra_key.size(0), self.world_size)
        self.rank_begin, self.rank_end = self.rank_idx[self.rank], self.rank_idx[self.rank + 1]

    def _rebuild_hash(self) -> None:
        ok = USE_HASH and self._bra_key.is_cuda and self._bra_key.size(0) > 0
        self.hashtable = CX.hash_build(self._bra_key, self.sorb) if ok else None  # values = positions in the sorted keys

    bra_key = property(lambda self: self._bra_key)
    wf_value = property(lambda self: self._wf_value)
    dtype = property(lambda self: self._wf_value.dtype)
    memory = property(lambda self: self._bra_key.numel() / 2**20)

    def to(self, device) -> None:
        self._bra_key, self._wf_value = self._bra_key.to(device=device), self._wf_value.to(device=device)
        self._rebuild_hash()

    def find(self, onv: Tensor) -> Tuple[Tensor, Tensor]:
        """(position of every onv in the sorted keys or -1, found mask)."""
        if self.hashtable is not None and onv.is_cuda:
            pos, found = CX.hash_lookup(self.hashtable, onv)
        else:
            pos, found = wavefunction_lut(self._bra_key, onv, self.sorb)
        return pos.to(onv.device), found.to(onv.device)

    def lookup(self, onv: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """(indices of onv found, indices not found, psi of the found ones) -- public_function.py:817-838."""
        pos, found = self.find(onv)
        every = torch.arange(onv.size(0), device=onv.device, dtype=torch.int64)
        return every[found], every[~found], self._wf_value[pos[found]]

    def index_value(self, begin: int, end: int) -> Tensor:
        if not self.sort:
            raise AssertionError("not-sorted does not support index-value")
        lo, hi = self.rank_begin + begin, self.rank_begin + end
        if hi > self.rank_end:
            raise AssertionError("Index date must be in the same rank")
        return self._wf_value[self.idx_sorted[lo:hi]]

    def clean_memory(self) -> None:
        self._bra_key = self._wf_value = None
        self.hashtable = None

    def __repr__(self) -> str:
        return (f"{type(self).__name__}(\n    bra-key shape: {tuple(self.bra_key.size())}\n"
                f"    wf-value shape: {self.wf_value.size(0)}\n    sorb: {self.sorb}\n    Memory: {self.memory:.3f} MiB\n)")


def ansatz_batch(func: Callable[[Tensor], Tensor], x: Tensor, batch: int, sorb: int, device, dtype) -> Tensor:
    """utils/public_function.py:934-960: uint8 onv -> +-1 -> func, in chunks of `batch` rows."""
    packed = x.dtype == torch.uint8
    if not packed and x.size(1) != sorb:
        raise AssertionError("expected +-1 rows of length sorb")
    feed = (lambda rows: onv_to_tensor(rows, sorb)) if packed else (lambda rows: rows)
    n = x.size(0)
    if batch == -1 or n == 0 or batch >= n:
        return func(feed(x)).to(dtype)
    out = torch.empty(n, device=device, dtype=dtype)
    lo = 0
    for hi in split_batch_idx(n, batch):
        out[lo:hi] = func(feed(x[lo:hi])).to(dtype).view(-1)
        lo = hi
    return out


# ---- spin-flip symmetry helpers (same results as utils/public_function.py:966-1036) --------------------------------
_DOUBLY = None  # [256] number of doubly occupied spatial orbitals among the four a byte of a packed determinant holds


# ---- walker-chunk sizing ------------------------------------------------------------------------------------------------------
def get_nbatch(sorb: int, n_sample: int, n_sd: int, Max_memory: float = 32, alpha: float = 0.25, device: Optional[torch.device] = None,
               use_sample: bool = False, dtype=torch.double, fused: Optional[str] = None, eps_sample: int = 0, kept_estimate: Optional[int] = None) -> int:
    """utils/public_function.py:162-261: how many walkers total_energy hands to one local_energy call.

    `fused=None` is the reference's arithmetic -- memory of the MATERIALISED path: the +-1 expansion of all nbatch x n_sd kets
    (_get_nbatch_simple, :183-206) or the Hij / psi(x') / comb_x / search arrays of the sample-space method
    (_get_nbatch_sample_space, :209-261), against min(Max_memory GiB, free device memory).

    The fused paths of pynqs_amd.energy allocate none of that, so sizing them with the reference's formula only multiplies launches
    (Fe2S2, SAMPLE_SPACE: the example's batch of 2048 walkers per call runs at a fraction of the rate of one 65 536-walker launch,
    bench.py extra "chunking").  `fused` names what will run and the estimate is of what THAT allocates per walker:
      "sample_space", "simple_rbm"  outputs only (E_loc, psi(x), partner sum): everything in one call, up to a 2^22-walker cap;
      "reduce"                      records (fixed + kept + eps_sample slots of 20 + 8 len bytes) plus, per distinct x', the +-1 row,
                                    the determinant and two de-duplication slots (every record counted as distinct: an upper bound);
                                    with draws 4 bytes per column for the row's float32 copy; kept_estimate defaults to n_sd / 64.
    The ansatz' own activation memory is bounded separately by fp_batch, as in the reference."""
    if device is None:
        device = torch.device("cpu")
    budget = float(Max_memory)
    if device.type != "cpu" and fused is None:
        torch.cuda.empty_cache()
        budget = min(torch.cuda.mem_get_info(device)[0] / (1 << 30), budget)
    elif device.type != "cpu":
        # (the fused paths ask once per total_energy call: emptying the allocator's cache there costs 0.26 ms per call and sends the call's
        # own buffers back to hipMalloc -- 17 % of a configs[1]-sized step; what the cache holds unused counts as free instead)
        cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        budget = min((torch.cuda.mem_get_info(device)[0] + max(cached, 0)) / (1 << 30), budget)
    if fused is None:
        if not use_sample:
            per = n_sd * sorb * 8 / (1 << 30) * 2
            return int(budget / per * alpha) if per * n_sample / budget >= alpha else n_sample
        bra_len = (sorb - 1) // 64 + 1
        if dtype not in (torch.double, torch.complex128):
            raise NotImplementedError
        cplx = dtype == torch.complex128
        a = max(alpha, 1)
        if n_sd * (2 + cplx + bra_len) * a <= n_sample:
            per = (((3 if cplx else 2) + bra_len) * n_sd + (12 if cplx else 8)) * 8 / (1 << 30)
        else:
            per = (n_sample * (2 if cplx else 1) + (10 if cplx else 6)) * 8 / (1 << 30)
        return min(n_sample, int(budget / per))
    if fused in ("sample_space", "simple_rbm"):
        return max(1, min(n_sample, 1 << 22, int(budget * alpha * (1 << 30) / 64)))
    if fused == "reduce":
        bra_len = (sorb - 1) // 64 + 1
        kept = int(kept_estimate) if kept_estimate is not None else max(64, n_sd // 64)
        nrec = kept + int(eps_sample) + 64
        per = nrec * (20 + 8 * bra_len) + nrec * (8 * sorb + 8 * bra_len + 2 * 8 * (2 if bra_len == 1 else 4))
        if eps_sample > 0:
            per += 4 * (n_sd + 17)   # the row's float32 copy of the semi-stochastic forms (ReduceFrontEnd.row_f32)
        return max(1, min(n_sample, int(budget * alpha * (1 << 30) / per)))
    raise ValueError(f"fused = {fused!r}")


class MemoryTrack:
    """utils/public_function.py:873-925: device memory before / after a block (allocated, peak), for sizing by observation."""

    def __init__(self, device) -> None:
        self.device = torch.device(device)
        self.before_memory = self.after_memory = self.before_max_memory = self.after_max_memory = 0.0

    def __enter__(self) -> "MemoryTrack":
        self.clean_memory_cache(self.device)
        if self.device.type == "cuda":
            torch.cuda.reset_peak_memory_stats(self.device)
        self.before_max_memory = self.get_max_memory(self.device)
        self.before_memory = self.get_current_memory(self.device)
        return self

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        self.after_max_memory = self.get_max_memory(self.device)
        self.clean_memory_cache(self.device)
        self.after_memory = self.get_current_memory(self.device)

    @property
    def used(self) -> float:
        """peak GiB allocated inside the block above what was allocated before it"""
        return self.after_max_memory - self.before_memory

    @staticmethod
    def get_max_memory(device) -> float:
        return torch.cuda.max_memory_allocated(device) / 2**30 if torch.device(device).type == "cuda" else 0.0

    @staticmethod
    def get_current_memory(device) -> float:
        return torch.cuda.memory_allocated(device) / 2**30 if torch.device(device).type == "cuda" else 0.0

    @staticmethod
    def clean_memory_cache(device) -> None:
        if torch.device(device).type == "cuda":
            torch.cuda.empty_cache()


def _doubly_table(device) -> Tensor:
    global _DOUBLY
    if _DOUBLY is None:
        _DOUBLY = torch.tensor([bin(b & (b >> 1) & 0x55).count("1") for b in range(256)], dtype=torch.int64)
    return _DOUBLY.to(device)


def spin_flip_onv(x: Tensor, sorb: int) -> Tensor:
    """The determinant with alpha and beta occupations exchanged (orbital 2k <-> 2k + 1): packed uint8 determinants or
    [n, sorb] occupation rows."""
    if x.dtype == torch.uint8:
        return ((x >> 1) & 0x55) | ((x & 0x55) << 1)
    assert x.size(1) == sorb
    return x.reshape(x.size(0), sorb // 2, 2).flip(-1).reshape(x.size(0), sorb)


def spin_flip_sign(x: Tensor, sorb: int) -> Tensor:
    """(-1)^(number of doubly occupied spatial orbitals): the sign the alpha <-> beta exchange gives a determinant.
    Packed uint8 determinants, or [n, sorb] rows with 1 = occupied, 0 = empty."""
    if x.dtype == torch.uint8:
        pairs = _doubly_table(x.device)[x.long()].sum(dim=-1)
    else:
        assert x.size(1) == sorb
        pairs = ((x[:, 0::2] == 1) & (x[:, 1::2] == 1)).sum(dim=1)
    return 1 - 2 * (pairs & 1)


class _SpinProjection:
    """eta = (-1)^(N // 2 - S) of the spin-projected local energies (set once per run with init(N, S))."""

    __slots__ = ("_eta",)

    def __init__(self) -> None:
        self._eta: Optional[int] = None

    def init(self, N: int, S: int) -> None:
        if not (isinstance(N, int) and isinstance(S, int)):
            raise AssertionError("N and S must be integers")
        self._eta = -1 if (N // 2 - S) % 2 else 1

    @property
    def eta(self) -> int:
        if self._eta is None:
            # inside a PyNQS process that imports pynqs_amd.energy (one-line change) the run initialises PyNQS' own
            # instance (utils/public_function.py:1017-1036): follow it instead of asking for a second init
            import sys

            host = getattr(sys.modules.get("utils.public_function"), "SpinProjection", None)
            if host is not None and host is not self:
                try:
                    return int(host.eta)
                except NotImplementedError:
                    pass
            raise NotImplementedError("SpinProjection.init(N, S) has not been called")
        return self._eta


SpinProjection = _SpinProjection()
